"""Host-side helpers of lancedb_amd/distributed.py that need no GPU: the communicator-id hand-off through a file
(the dependency-free channel of the one-process-per-GPU launcher) and the thread driver of a loopback group."""
import os
import threading
import time

import pytest

from lancedb_amd import _abi, distributed


def test_id_file_is_refused_by_rank0_when_it_already_exists(tmp_path):
    """A file left by an earlier launch would hand the other ranks a dead id: rank 0 must not start over it
    (checked before any library call, so this runs without a device)."""
    p = tmp_path / "comm.id"
    p.write_bytes(b"\0" * _abi.COMM_ID_BYTES)
    with pytest.raises(FileExistsError):
        distributed.exchange_id_via_file(str(p), rank=0)


def test_other_ranks_wait_for_a_complete_id(tmp_path):
    p = tmp_path / "comm.id"
    uid = bytes((7 * i + 3) % 256 for i in range(_abi.COMM_ID_BYTES))

    def writer():
        time.sleep(0.15)
        p.write_bytes(uid[:17])  # a torn write: too short to be taken
        time.sleep(0.15)
        tmp = str(p) + ".tmp"
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, str(p))

    t = threading.Thread(target=writer)
    t.start()
    got = distributed.exchange_id_via_file(str(p), rank=3, timeout_s=5.0)
    t.join()
    assert got == uid


def test_other_ranks_time_out_without_an_id(tmp_path):
    with pytest.raises(TimeoutError):
        distributed.exchange_id_via_file(str(tmp_path / "never"), rank=1, timeout_s=0.2)


def test_run_ranks_returns_in_rank_order_and_reraises_the_first_error():
    order = []

    def mk(i, delay):
        def fn():
            time.sleep(delay)
            order.append(i)
            return i * i
        return fn

    assert distributed.run_ranks([mk(0, 0.05), mk(1, 0.0), mk(2, 0.02)]) == [0, 1, 4]
    assert sorted(order) == [0, 1, 2] and order[0] == 1  # they ran concurrently

    def boom():
        raise ValueError("rank 1 failed")

    with pytest.raises(ValueError, match="rank 1 failed"):
        distributed.run_ranks([mk(0, 0.0), boom, mk(2, 0.0)])
