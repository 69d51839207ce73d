"""The Philox contract of include/dnaf.h says: draw n of RNG row r is rocRAND's rocrand_state_philox4x32_10 sequence for
(seed, subsequence = r, offset = n).  tools/rocrand_check.hip asks rocRAND itself (rocrand_init + rocrand(), header-only,
/opt/rocm/include/rocrand); here its answers are compared with the oracle's draw -- word n & 3 of
Philox4x32-10(counter = (n >> 2, 0, r_lo, r_hi), key = seed) -- on the host (no GPU needed: the engine is __host__
__device__) and, on the GPU box, on the device."""
import os
import shutil
import subprocess

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(REPO, "tools", "rocrand_check")


def _cases():
    rs = np.random.RandomState(2026)
    cases = [(12345, 0, 0), (12345, 0, 1), (12345, 0, 3), (12345, 0, 4), (12345, 1, 0), (0, 0, 0), (2**64 - 1, 2**64 - 1, 2**31 - 1),
             (0xFEDCBA9876543210, (1 << 32) - 4, 39999), (77, 1 << 32, 2), (77, (1 << 40) + 5, 123456)]
    for _ in range(200):
        cases.append((int(rs.randint(0, 2**63)) * 2 + int(rs.randint(0, 2)), int(rs.randint(0, 2**62)), int(rs.randint(0, 2**31))))
    return cases


def _run_tool():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(TOOL) or os.path.getmtime(TOOL) < os.path.getmtime(TOOL + ".hip"):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-o", TOOL, TOOL + ".hip"], stderr=subprocess.DEVNULL)
    cases = _cases()
    p = subprocess.run([TOOL], input="".join("%d %d %d\n" % c for c in cases), capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    rows = [l.split() for l in p.stdout.splitlines()]
    assert len(rows) == len(cases)
    return cases, rows


def _oracle_draw(orc, seed, row, n):
    ctr = np.array([n >> 2, 0, row & 0xFFFFFFFF, row >> 32], dtype=np.uint32)
    key = np.array([seed & 0xFFFFFFFF, seed >> 32], dtype=np.uint32)
    return int(orc.philox4x32_10(ctr, key)[n & 3])


def test_rocrand_host_engine_is_the_oracle_draw(oracle):
    cases, rows = _run_tool()
    for (seed, row, n), r in zip(cases, rows):
        assert int(r[0]) == _oracle_draw(oracle, seed, row, n), (seed, row, n)


@pytest.mark.gpu
def test_rocrand_device_sequence_is_the_oracle_draw(oracle):
    cases, rows = _run_tool()
    assert all(r[1] != "-" for r in rows), "no GPU seen by tools/rocrand_check"
    for (seed, row, n), r in zip(cases, rows):
        want = _oracle_draw(oracle, seed, row, n)
        assert int(r[1]) == want and int(r[0]) == want, (seed, row, n)
