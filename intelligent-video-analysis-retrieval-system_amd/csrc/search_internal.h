// Internal interface between search.hip (index object, selection, re-score) and search_scanq.hip (the large-query-batch
// candidate scan).  Not part of the C ABI.
#pragma once
#include "ivr_common.h"

// One launch of the large-query candidate scan: every stored row against every query of the batch on the bf16 MFMA,
// reduced on the fly to one maximum per (query, 16-row tile) and one per (query, 128-row block).  The index is streamed
// from HBM once per launch whatever the number of queries ("large query batches").
//   data16 : bf16 scan copy of the rows, [row tile of 16][pieces][64 lanes x 16 B]; allocation padded to 256 rows
//   q16    : the queries in the same layout (bf16, rounded to nearest), padded with zero rows to a multiple of 256 queries
//   pieces : 1 KiB pieces per 16-row tile (even)
//   tmax   : [padded queries][tstride] maximum of each 16-row tile; bmax: [padded queries][bstride] maximum of each 128-row block
struct ScanQArgs {
    const uint4 *data16;
    const uint4 *q16;
    int pieces;
    int qblocks;          // padded queries / 256
    int64_t ntotal;       // stored rows; rows >= ntotal are masked out of the maxima
    int64_t nblocks;      // ceil(ntotal / 256)
    float *tmax;
    int64_t tstride;
    float *bmax;
    int64_t bstride;
};

int ivr_launch_scanq(ivr_ctx *ctx, const ScanQArgs &a, hipStream_t s);
