"""Two-rank rehearsal of bench.py's control flow on ONE GPU (gloo collectives through the host, both ranks on cuda:0):
the launch contract (torch.distributed.run, env rendezvous), rank-0-only measurement legs that must not issue
collectives, and the single JSON line.  The reported numbers of a real multi-GPU run use RCCL; this only proves the
flow terminates and the line is well-formed."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("graph", ["off", "on", "on-capture-fails-on-rank-1"])
def test_two_rank_bench_flow(graph):
    env = dict(os.environ, PN_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    fail = graph.startswith("on-")
    if fail:  # a capture that throws on ONE rank must send BOTH ranks to eager launches through the same collective
        env["PN_BENCH_FAIL_CAPTURE_RANK"] = "1"
        graph = "on"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
           "--warmup", "1", "--global-batch", "256", "--samples", "32", "--height", "64", "--width", "128", "--graph",
           graph]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "strong"
    assert out["config"]["rays_per_gpu"] == 128 and out["value"] > 0
    assert out["roofline"]["bound"] in ("mfma", "hbm") and 0 < out["roofline"]["frac"] < 1
    assert out["roofline"]["kernel"].startswith("k_chain_")
    assert "cpu_baseline" not in out  # rank 0 at N = 1 only
    # the audit record of the data-parallel step: both ranks took part, each with its shard and its own loss, and the
    # gradient block is the same on both after the all-reduce
    audit = out["config"]["data_parallel_audit"]
    assert audit["rays_per_rank"] == [128, 128] and len(audit["last_loss_per_rank"]) == 2
    assert audit["last_loss_per_rank"][0] != audit["last_loss_per_rank"][1]
    assert audit["allreduced_grad_l2"] > 0 and audit["allreduced_grad_l2_spread_over_ranks"] == 0.0
    if fail:
        assert out["config"]["launch"].startswith("eager (graph capture failed")
        assert out["config"]["replay_check"]["ok_all_ranks"] is False and out["config"]["replay_check"]["ok_this_rank"] is True
    elif graph == "on":  # the captured step was checked against an eager step on the same batch before it was timed
        assert out["config"]["launch"] == "hip-graph replay"
        assert out["config"]["replay_check"]["ok_all_ranks"] is True
        assert out["config"]["replay_check"]["max_grad_diff_over_max_grad"] <= 1e-6
    else:
        assert out["config"]["launch"] == "eager" and out["config"]["replay_check"] is None


def test_two_rank_render_matches_single_rank():
    """Sharded full-image inference: chunks dealt round-robin to two ranks, gathered, re-ordered."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29519", os.path.join(ROOT, "tests", "_render_dist_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    assert "RENDER_DIST_OK" in res.stdout


def test_ddp_wrapped_module_trains():
    """'Lightning DDP wraps the module as usual' — DistributedDataParallel + torch Adam over the flat-view
    parameters: parameters stay flat, the packed weights follow optimizer.step(), replicas stay bit-equal."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29521", os.path.join(ROOT, "tests", "_ddp_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
    assert "DDP_OK" in res.stdout
