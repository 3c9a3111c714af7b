"""world_size > 1 on the CPU (gloo): contiguous patch sharding, padded slabs, all-gather."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,problems", [(2, 1), (3, 1), (2, 2)])
def test_sharded_all_gather_matches_single_process(world, problems):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SLOD_TEST_PROBLEMS=str(problems), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py")],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "DIST_OK" in outs[0]


def test_gather_pieces_tile_the_padded_slab():
    """slod_gather_piece: the pieces of slod_plan_execute_allgather cover a rank's padded slab exactly
    once, in order, identically on every rank (host-only index calculus; the exchange itself needs
    RCCL and GPUs: tests/test_gpu_lod_system.py::test_execute_allgather_single_rank)."""
    import slod_amd
    for total, world in ((1024, 2), (1024, 3), (17, 2), (5, 8)):
        ppr = max(slod_amd.partition(total, world, r)[1] - slod_amd.partition(total, world, r)[0] for r in range(world))
        for n_pieces in (1, 2, 3, 7, ppr, ppr + 3):
            covered = []
            for i in range(n_pieces):
                f, c = slod_amd.gather_piece(ppr, n_pieces, i)
                assert f + c <= ppr
                covered += list(range(f, f + c))
            assert covered == list(range(ppr)), (total, world, n_pieces)
