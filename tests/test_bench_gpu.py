"""bench.py's contract on the GPU box: the JSON line, and the multi-GPU forms -- stand-alone (`--gpus N` without a
launcher: N contexts on N host threads, no RCCL) and under torchrun (one process per GPU, gloo barrier).  On the
one-GPU box the ranks share GPU 0 (DNAF_BENCH_SHARE_GPU=1, a rehearsal knob the driver never sets)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--samples", "2000", "--tile-rows", "20000", "--table-rows", "40000", "--steps", "3", "--warmup", "1",
         "--cpu-seconds", "0", "--bgzf-tiles", "0"]


def _env(**kw):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(kw)
    return env


def _last_json(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    p = subprocess.run([sys.executable, "bench.py"] + SMALL, cwd=REPO, env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["unit"] == "SNP-rows/s" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert abs(d["value"] - 3 * 20000 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] is None and "workload" in r["traffic_source"]  # the committed PMC file is for the default workload


def test_mt19937_mode_line():
    """--rng mt19937: tiles are whole multiples of the stream's row quantum (156 / gcd(S, 156) rows per wave)."""
    args = ["--rng", "mt19937", "--samples", "2000", "--tile-rows", "20000", "--table-rows", "40000", "--steps", "2", "--warmup", "1",
            "--cpu-seconds", "0", "--bgzf-tiles", "0"]
    p = subprocess.run([sys.executable, "bench.py"] + args, cwd=REPO, env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _last_json(p.stdout)
    assert d["config"]["rng"].startswith("mt19937") and d["config"]["tile_rows"] % 39 == 0 and d["value"] > 0
    assert "mt_rowgen_kernel" in d["roofline"]["kernel"]


def test_gpus_2_stands_alone_without_a_launcher():
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + SMALL, cwd=REPO, env=_env(DNAF_BENCH_SHARE_GPU="1"),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 2 and len(d["config"]["per_gpu_wall_s"]) == 2
    assert "host thread" in d["config"]["launcher"]
    # whole-job rows over the slowest rank's time
    assert abs(d["value"] - 2 * 3 * 20000 / max(d["config"]["per_gpu_wall_s"])) < 1e-6 * d["value"]


def test_more_gpus_than_the_node_has_is_an_error():
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "8"] + SMALL, cwd=REPO, env=_env(), capture_output=True, text=True, timeout=600)
    from dna_factory_amd import capi

    if capi.device_count() >= 8:
        pytest.skip("this node has 8 GPUs")
    assert p.returncode != 0 and not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_torchrun_two_ranks_use_gloo_only():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2"] + SMALL
    p = subprocess.run(cmd, cwd=REPO, env=_env(DNAF_BENCH_SHARE_GPU="1"), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 2 and "gloo" in d["config"]["launcher"] and len(d["config"]["per_gpu_wall_s"]) == 2
