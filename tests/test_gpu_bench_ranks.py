"""bench.py --gpus N without a launcher: the script starts its N ranks itself (torch.distributed.run on 127.0.0.1), every rank joins the library's
communicator, and the JSON line says what the LIBRARY saw.  Run here with two ranks on ONE device (LOCAL_RANK % device_count; the harness's own
exchanges go over gloo, the per-iteration collective over the library's peer buffers -- RCCL wants one GPU per rank), so that the first N > 1 run on
real hardware is not the first run of this code path."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(out):
    rows = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert rows, out.stdout[-2000:] + out.stderr[-4000:]
    return json.loads(rows[-1])


def test_bench_gpus_2_spawns_two_ranks_that_share_one_window():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--ipc", "--steps", "4", "--warmup", "1", "--no-cpu", "--no-cfg3"],
                         env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    r = _line(out)
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["steps"] == 4, r
    assert r["value"] > 0 and "note" not in r["config"], r["config"]          # (a note = the communicator could not be set up and the ranks fell back to replicas)
    ph = r["phases_per_rank"]["per_rank"]
    assert len(ph) == 2 and all(len(row) == 4 and row[0] > 0 and row[2] > 0 and row[3] > 0 for row in ph), ph      # sweep, gather, COLLECTIVE, step: every rank timed all four
    cm = r["communicator"]
    assert cm["ranks_seen_by_the_library"] == 2 and [row[:3] for row in cm["per_rank"]] == [[0, 2, 3], [1, 2, 3]], cm
    assert 0 < cm["message_bytes_per_peer_rank0"] < cm["full_set_bytes"]
    assert r["replicas"]["value"] > 0


def test_bench_gpus_2_replicas():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--replicas", "--steps", "4", "--warmup", "1", "--no-cpu", "--no-cfg3", "--no-events"],
                         env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    r = _line(out)
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["value"] > 0, r
