"""CPU-only: the JSON line bench.py prints (kept per round under profiles/) carries what the driver and the judge read — metric / value /
unit / timing fields, the `roofline` and `cpu_baseline` objects, a `config.workload`, no model keys — and its numbers are consistent
with each other.  Also: bench.py does not run without a GPU (no CPU fallback), and the oracle is only its checker / baseline."""
import glob
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_line(path):
    with open(path) as f:
        lines = [ln for ln in f.read().splitlines() if ln.strip().startswith("{")]
    return json.loads(lines[-1])


def _latest_default_line():
    files = sorted(p for p in glob.glob(os.path.join(ROOT, "profiles", "bench_r*.json")) if "config" not in os.path.basename(p))
    assert files, "no bench line under profiles/"
    return files[-1], _last_line(files[-1])


def test_default_bench_line_contract():
    path, d = _latest_default_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, (path, k)
    assert d["unit"] == "Mpix/s" and d["higher_is_better"] is True and d["n_gpus"] == 1 and d["vs_baseline"] is None
    assert "4096" in d["metric"] and "7x7 SAD" in d["metric"]
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["dtype"] == "u8" and "synthetic" in d["data"]
    # value = output pixels per second of the whole step
    px = 4090 * 4090
    assert abs(d["value"] - px / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.02
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    us = r["avg_us_per_launch"]["bm_sad_u8"]
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (us * 1e-6) / 1e9) / r["achieved"] < 1e-6
    assert r["algorithmic_bytes_per_launch"] == 4 * 4096 * 4096 + 4 * 4224 * 4096 + 12 * 4090 * 4090
    # the kernel fits inside the step — a step is ONE launch since round 4, and the sampled steps (every 4th) carry the two event records
    # around the kernel (~1.5 us), so the event average may exceed the all-steps wall average by that much
    assert us * 1e-3 <= d["ms_per_step"] * 1.01
    assert r["traffic"] is None or r["traffic"] >= 0.9 * r["algorithmic_bytes_per_launch"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["unit"] == "Mpix/s" and c["result_identical_to_oracle"] is True
    for e in d.get("extra", []):
        assert "name" in e
        if e.get("roofline_frac") is not None:
            assert 0 < e["roofline_frac"] < 1 and e["algorithmic_bytes"] > 0 and e.get("hot_us", 1) > 0


@pytest.mark.parametrize("name", ["config4", "config5"])
def test_config_mode_lines(name):
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "bench_r*_%s.json" % name)))
    assert files
    d = _last_line(files[-1])
    assert d["unit"] == "Mpix/s" and d["value"] > 0 and d["n_gpus"] == 1 and "workload" in d["config"]
    assert ("16384" in d["metric"]) if name == "config4" else ("32768" in d["metric"])


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert not any(ln.strip().startswith("{") for ln in p.stdout.splitlines())        # no JSON line from a CPU path


def test_bench_gpus_2_launches_its_own_ranks():
    """`python bench.py --gpus 2` as typed (no RANK in the environment) must re-execute itself under torch.distributed.run: both ranks
    come up with RANK / WORLD_SIZE / LOCAL_RANK set and — on this GPU-less host — stop at the 'no GPU visible' line, each for itself."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert p.returncode != 0
    err = p.stderr
    # (the launcher stops the surviving rank as soon as the first one exits: one line is guaranteed, the second usually makes it)
    seen = re.findall(r"rank ([01]) of 2 \(local rank ([01])\): no GPU visible", err)
    assert seen and all(a == b for a, b in seen), err[-2000:]
    assert "must be launched with" not in err
    assert not any(ln.strip().startswith("{") for ln in p.stdout.splitlines())
    # a launch whose world size contradicts --gpus is refused with a message that says how to launch
    env2 = dict(env, RANK="0", WORLD_SIZE="4", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env2)
    assert p.returncode != 0 and "WORLD_SIZE is 4" in p.stderr


def test_halo_fetcher_decisions_are_collective():
    """bench.py's HaloFetcher: world 2 over gloo on the CPU (the engine communicator cannot exist without a GPU, so both ranks must
    agree on the torch.distributed exchange and return the right rows)."""
    script = os.path.join(ROOT, "tests", "_halo_fetcher_worker.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    import socket
    with socket.socket() as sock:                   # a free port, as bench.py's own launcher picks one (a fixed one can collide on a shared host)
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), script], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    assert p.stdout.count("halo fetcher ok") == 2, p.stdout


def test_oracle_is_only_the_checker_in_bench():
    src = open(os.path.join(ROOT, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"\boracle\b", src)]
    assert uses
    # every mention sits in the docstrings / the cpu_baseline leg, none in the timed step
    step = src[src.index("def step():"):src.index("def barrier():")]
    assert "oracle" not in step
