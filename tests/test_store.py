"""The TCP rendezvous store (distributedpytorch_amd/store.py): what carries the RCCL unique id when a job runs
without torch.distributed (comm.init_native_group).  Reference behaviour replaced: c10d's TCPStore behind
``init_method="tcp://..."`` (multinode-DP/multinodeDP.py:161)."""
import threading
import time

import pytest

from distributedpytorch_amd.store import TCPKVStore, parse_tcp_url


def test_set_get_add_and_objects_between_clients():
    master = TCPKVStore("127.0.0.1", 0, is_master=True, timeout_s=10)
    other = TCPKVStore("127.0.0.1", master.port, is_master=False, timeout_s=10)
    try:
        master.set("uid", bytes(range(128)))
        assert other.get("uid") == bytes(range(128))
        assert other.add("n", 2) == 2 and master.add("n", 3) == 5
        other.set_object("o", {"rank": 1, "vals": [1.5, 2.5]})
        assert master.get_object("o") == {"rank": 1, "vals": [1.5, 2.5]}
        master.delete("uid")
        with pytest.raises(TimeoutError):
            other.get("uid", timeout_s=0.2)
    finally:
        other.close()
        master.close()


def test_get_blocks_until_another_client_sets_the_key_and_barrier_releases_everyone():
    master = TCPKVStore("127.0.0.1", 0, is_master=True, timeout_s=10)
    clients = [TCPKVStore("127.0.0.1", master.port, is_master=False, timeout_s=10) for _ in range(3)]
    got, order = {}, []

    def waiter(i):
        got[i] = clients[i].get("late")
        clients[i].barrier("b0", 4)
        order.append(i)

    threads = [threading.Thread(target=waiter, args=(i,)) for i in range(3)]
    for t in threads:
        t.start()
    time.sleep(0.2)
    assert not got
    master.set("late", b"x")
    time.sleep(0.2)
    assert got == {0: b"x", 1: b"x", 2: b"x"} and not order          # all three are held by the barrier
    master.barrier("b0", 4)
    for t in threads:
        t.join(5)
    assert sorted(order) == [0, 1, 2]
    for c in clients:
        c.close()
    master.close()


def test_parse_tcp_url():
    assert parse_tcp_url("tcp://10.0.0.3:8885") == ("10.0.0.3", 8885)
    assert parse_tcp_url("127.0.0.1:29500") == ("127.0.0.1", 29500)


def test_values_with_a_reader_count_leave_the_store_and_barriers_can_be_reused():
    master = TCPKVStore("127.0.0.1", 0, is_master=True, timeout_s=10)
    other = TCPKVStore("127.0.0.1", master.port, is_master=False, timeout_s=10)
    try:
        base = master.num_keys()
        master.set_object("obj/bcast/1", {"w": [1, 2, 3]}, readers=2)
        assert other.get_object("obj/bcast/1") == {"w": [1, 2, 3]}
        assert master.num_keys() == base + 1                    # one reader still to come
        assert master.get_object("obj/bcast/1") == {"w": [1, 2, 3]}
        assert master.num_keys() == base                        # the last reader took it along
        with pytest.raises(TimeoutError):
            other.get("obj/bcast/1", timeout_s=0.2)
        for _ in range(3):                                      # the same name, three rounds
            t = threading.Thread(target=lambda: other.barrier("again", 2))
            t.start()
            master.barrier("again", 2)
            t.join(5)
            assert not t.is_alive()
        assert master.num_keys() <= base + 2                    # only the last round's two keys are left
    finally:
        other.close()
        master.close()


def test_a_client_without_the_jobs_token_is_dropped(monkeypatch):
    monkeypatch.setenv("DPA_STORE_TOKEN", "s3cret")
    master = TCPKVStore("127.0.0.1", 0, is_master=True, timeout_s=5)
    good = TCPKVStore("127.0.0.1", master.port, is_master=False, timeout_s=5)
    good.set("k", b"v")
    monkeypatch.setenv("DPA_STORE_TOKEN", "guess")
    bad = TCPKVStore("127.0.0.1", master.port, is_master=False, timeout_s=5)
    try:
        with pytest.raises((ConnectionError, OSError)):
            bad.set("planted", b"x")
        with pytest.raises(TimeoutError):
            good.get("planted", timeout_s=0.2)
        assert good.get("k") == b"v"
    finally:
        bad.close(); good.close(); master.close()


def test_a_timed_out_get_raises_the_stores_own_error():
    master = TCPKVStore("127.0.0.1", 0, is_master=True, timeout_s=1.0)
    try:
        t0 = time.monotonic()
        with pytest.raises(TimeoutError) as e:
            master.get("never")                                 # request timeout == the client's default: no socket.timeout
        assert "never" in str(e.value) and time.monotonic() - t0 < 10
    finally:
        master.close()
