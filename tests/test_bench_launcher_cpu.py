"""CPU suite: `python bench.py --gpus N` from a plain shell must start its own ranks.
The launcher branch is exercised with a stand-in rank script (gloo, world_size 2): bench.py itself needs a GPU."""
import json
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

import bench


def test_launcher_command_shape():
    cmd = bench.launcher_command(4, "/x/bench.py", ["--gpus", "4", "--steps", "3"], port=29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == ["/x/bench.py", "--gpus", "4", "--steps", "3"]


RANK_SCRIPT = textwrap.dedent("""
    import json, os, sys
    import torch, torch.distributed as dist
    assert os.environ["MASTER_ADDR"] == "127.0.0.1"
    dist.init_process_group(os.environ.get("IVR_DIST_BACKEND", "nccl"), rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    t = torch.tensor([float(os.environ["RANK"]) + 1.0])
    dist.all_reduce(t)
    print("noise from rank", os.environ["RANK"])
    if dist.get_rank() == 0:
        print(json.dumps({"n_gpus": dist.get_world_size(), "sum": float(t.item()), "argv": sys.argv[1:]}))
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(int(os.environ.get("FAKE_RC", "0")))
""")


def _run_self_launch(tmp_path, extra_env=None):
    script = tmp_path / "fake_rank.py"
    script.write_text(RANK_SCRIPT)
    driver = tmp_path / "driver.py"
    driver.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench
        sys.exit(bench.self_launch(2, {str(script)!r}, ["--gpus", "2", "--steps", "3"], visible_gpus=0))
    """))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("IVR_DIST_BACKEND", None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, str(driver)], env=env, capture_output=True, text=True, timeout=300)


def test_self_launch_relays_rank0_json(tmp_path):
    p = _run_self_launch(tmp_path)
    assert p.returncode == 0, p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout                      # ONE JSON line on stdout, everything else on stderr
    out = json.loads(lines[0])
    assert out == {"n_gpus": 2, "sum": 3.0, "argv": ["--gpus", "2", "--steps", "3"]}
    assert "backend gloo (rehearsal)" in p.stderr and "noise from rank" in p.stderr


def test_self_launch_propagates_failure(tmp_path):
    p = _run_self_launch(tmp_path, {"FAKE_RC": "3"})
    assert p.returncode != 0
