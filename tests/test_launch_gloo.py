"""the control plane of a multi-rank run on CPU: two processes under gloo (what torchrun starts for `bench.py --gpus 2`) agree
on the communicator id, the barrier and the max-over-ranks timing; the ownership arithmetic covers every read and bucket once."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from disco_amd import launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json
sys.path.insert(0, %r)
from disco_amd import launch
cp = launch.ControlPlane("gloo")
uid = cp.broadcast_unique_id(lambda: bytes(range(128)))
cp.barrier()
t = cp.max_over_ranks(1.0 + cp.rank)
lo, hi = launch.owner_range(1000, cp.rank, cp.world)
print(json.dumps({"rank": cp.rank, "world": cp.world, "uid_ok": uid == bytes(range(128)), "t": t, "range": [lo, hi]}), flush=True)
cp.close()
"""


@pytest.mark.parametrize("world", [2, 3])
def test_control_plane_over_gloo(world, tmp_path):
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-2000:]
        outs.append(__import__("json").loads(o.strip().splitlines()[-1]))
    assert sorted(o["rank"] for o in outs) == list(range(world))
    assert all(o["uid_ok"] and o["world"] == world and o["t"] == float(world) for o in outs)
    ranges = sorted(tuple(o["range"]) for o in outs)
    assert ranges[0][0] == 0 and ranges[-1][1] == 1000 and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))


@pytest.mark.parametrize("n,world", [(0, 2), (1, 3), (63, 2), (64, 2), (65, 2), (1000, 8), (50_000_000, 8), (12345677, 7)])
def test_ownership_covers_everything_once(n, world):
    r = [launch.owner_range(n, k, world) for k in range(world)]
    assert r[0][0] == 0 and r[-1][1] == n
    assert all(a[1] == b[0] and a[0] <= a[1] for a, b in zip(r, r[1:]))
    per = {hi - lo for lo, hi in r if hi - lo and hi != n}
    assert all(p % 64 == 0 for p in per) and len(per) <= 1  # full ranges are equal and bitmap-word aligned
    for logt in (10, 17, 27):
        b = [launch.bucket_range(logt, k, world) for k in range(world)]
        assert b[0][0] == 0 and b[-1][1] == 1 << logt and all(x[1] == y[0] for x, y in zip(b, b[1:]))
        probe = np.unique(np.concatenate([np.array([lo, max(lo, hi - 1)]) for lo, hi in b if hi > lo]))
        own = (probe.astype(object) * world) >> logt
        for v, o in zip(probe, own):
            assert b[int(o)][0] <= v < b[int(o)][1]
