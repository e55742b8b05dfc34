"""bench.py as its own launcher (CPU): ``python bench.py --gpus N`` without a launcher's environment must start N rank processes as
children -- before this process touches a GPU, never by exec -- relay rank 0's JSON line, and fail when any rank fails."""
import json
import os
import sys
import textwrap

import pytest

import bench


def _script(tmp_path, body):
    p = tmp_path / "ranks.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_spawn_ranks_relays_rank0_line(tmp_path, capfd):
    script = _script(tmp_path, """
        import json, os, sys
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
        print(f"noise from rank {rank}")
        if rank == 0:
            print(json.dumps({"metric": "PDHG iterations/sec", "n_gpus": world, "argv": sys.argv[1:]}))
    """)
    rc = bench.spawn_ranks(2, ["--steps", "7", "--n", "5"], script=script)
    out, err = capfd.readouterr()
    assert rc == 0
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1                                   # exactly the JSON line on stdout; everything else goes to stderr
    got = json.loads(lines[0])
    assert got["n_gpus"] == 2 and got["argv"] == ["--steps", "7", "--n", "5"]      # (torch.distributed.run would claim --n)
    assert "noise from rank 1" in err


def test_spawn_ranks_fails_when_a_rank_fails(tmp_path, capfd):
    script = _script(tmp_path, """
        import json, os, sys, time
        if int(os.environ["RANK"]) == 1:
            sys.exit(3)
        time.sleep(60)                                       # (rank 0 would be stuck in a collective: the launcher must stop it)
        print(json.dumps({"metric": "x"}))
    """)
    rc = bench.spawn_ranks(2, [], script=script)
    out, err = capfd.readouterr()
    assert rc != 0 and out.strip() == ""                     # no result line from a failed job
    assert "failed" in err


def test_spawn_ranks_relays_a_measured_line_when_ranks_hang_afterwards(tmp_path, capfd):
    """a rank stuck in a communicator's teardown after the result line must not cost the number"""
    script = _script(tmp_path, """
        import json, os, sys, time
        if int(os.environ["RANK"]) == 0:
            print(json.dumps({"metric": "PDHG iterations/sec", "value": 1.5}), flush=True)
        time.sleep(600)
    """)
    rc = bench.spawn_ranks(2, [], script=script, grace=1.0)
    out, err = capfd.readouterr()
    line = json.loads(out.strip())
    assert rc == 0 and line["value"] == 1.5 and line["degraded"] is True and "still running" in line["degraded_reason"]
    assert "stopping the ranks" in err


def test_spawn_ranks_gives_up_on_ranks_that_hang_before_a_line(tmp_path, capfd):
    script = _script(tmp_path, "import time\ntime.sleep(600)\n")
    rc = bench.spawn_ranks(2, [], script=script, launch_timeout=1.5)
    out, err = capfd.readouterr()
    assert rc != 0 and out.strip() == "" and "no result line" in err


def test_spawn_ranks_keeps_the_line_when_a_rank_fails_after_it(tmp_path, capfd):
    script = _script(tmp_path, """
        import json, os, sys, time
        if int(os.environ["RANK"]) == 0:
            print(json.dumps({"metric": "PDHG iterations/sec", "value": 2.5}), flush=True)
            time.sleep(600)
        time.sleep(1.0)
        sys.exit(7)
    """)
    rc = bench.spawn_ranks(2, [], script=script)
    out, err = capfd.readouterr()
    line = json.loads(out.strip())
    assert rc == 0 and line["value"] == 2.5 and "measured before the failure" in err
    assert line["degraded"] is True and "exited with code 7" in line["degraded_reason"]        # (a salvaged line says so itself)


def test_spawn_ranks_hosts_the_rendezvous_store(tmp_path, capfd):
    """no probe-and-release of a port; the ranks join a store the launcher keeps open"""
    script = _script(tmp_path, """
        import json, os
        import torch.distributed as dist
        assert os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"
        dist.init_process_group("gloo")
        import torch
        t = torch.tensor([float(dist.get_rank() + 1)])
        dist.all_reduce(t)
        if dist.get_rank() == 0:
            print(json.dumps({"metric": "PDHG iterations/sec", "value": float(t)}), flush=True)
        dist.destroy_process_group()
    """)
    rc = bench.spawn_ranks(2, [], script=script)
    out, _ = capfd.readouterr()
    assert rc == 0 and json.loads(out.strip())["value"] == 3.0


def test_spawn_ranks_fails_without_a_result_line(tmp_path, capfd):
    script = _script(tmp_path, "print('nothing useful')\n")
    assert bench.spawn_ranks(2, [], script=script) == 1
    out, _ = capfd.readouterr()
    assert out.strip() == ""


def test_main_becomes_the_launcher_only_without_a_launcher_environment(monkeypatch):
    calls = []
    monkeypatch.setattr(bench, "spawn_ranks", lambda gpus, argv, **kw: calls.append((gpus, list(argv))) or 0)
    monkeypatch.setattr(bench, "_imports", lambda: pytest.fail("the launcher process must not load the GPU stack"))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.main(["--gpus", "4", "--steps", "20", "--warmup", "5"]) == 0
    assert calls == [(4, ["--gpus", "4", "--steps", "20", "--warmup", "5"])]
    # under a launcher the rank count must match --gpus: a mismatch is an error, not a silent n_gpus = 1
    monkeypatch.setenv("WORLD_SIZE", "3")
    assert bench.main(["--gpus", "2"]) == 2
    monkeypatch.delenv("WORLD_SIZE")
    assert bench.main(["--gpus", "0"]) == 2
    assert len(calls) == 1


def test_gpus_defaults_to_the_launchers_world_size(monkeypatch):
    seen = {}

    def stop():
        raise RuntimeError("reached the compute part")
    monkeypatch.setattr(bench, "_imports", stop)
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(RuntimeError, match="compute part"):          # no --gpus: WORLD_SIZE decides, no mismatch error
        bench.main(["--steps", "1"])
    monkeypatch.delenv("WORLD_SIZE")
    with pytest.raises(RuntimeError, match="compute part"):          # neither: one rank, in this process
        bench.main(["--steps", "1"])
    assert seen == {}


# ---------------------------------------------------------------------------------------------------
# the library-communicator phase of a multi-rank run (first contact with a second RCCL communicator): whatever it does, the line that
# was measured on the torch.distributed loop before it is what comes out
# ---------------------------------------------------------------------------------------------------
class _Eng:
    lib_comm = None
    lib_comm_log = ["log line"]

    def __init__(self, behaviour):
        self.behaviour = behaviour

    def enable_library_comm(self, timeout, cross_check):
        assert cross_check
        if self.behaviour == "raise":
            raise RuntimeError("boom")
        self.lib_comm = self.behaviour == "on"
        return self.behaviour == "on"


def _line():
    return {"metric": "PDHG iterations/sec", "value": 100.0, "ms_per_step": 10.0, "config": {"workload": "w"}, "timing": {"elapsed_s": 0.2}}


def _region(norm):
    return lambda: dict(norm_elapsed=norm, elapsed=norm, checks_in=0, restarts_in=0, check_s=0.001)


def test_library_phase_outcomes():
    for behaviour, norm2, want_path, want_value in (("off", None, "declined", 100.0), ("raise", None, "failed: RuntimeError", 100.0),
                                                    ("on", 0.1, "library RCCL communicator", 200.0), ("on", 0.4, "slower here", 100.0)):
        out, ex, eng = _line(), {"path": "torch.distributed loop"}, _Eng(behaviour)
        bench.library_phase(out, ex, eng, 20, _region(norm2), 0.2, rank=0, limit=0.5, slack=0.5)
        assert want_path in ex["path"], (behaviour, ex)
        assert out["value"] == want_value
        if behaviour == "on":
            assert ex["torch_loop_value"] == 100.0 and ex["library_value"] == round(20 / norm2, 3) and ex["log"] == ["log line"]
            assert eng.lib_comm == (norm2 < 0.2)              # the slower driver is switched off again on every rank
        if behaviour == "raise":
            assert eng.lib_comm is False


def test_a_hung_library_phase_still_prints_the_measured_line(tmp_path):
    """a rank stuck inside ncclCommInitRank: the watchdog prints the torch.distributed line from rank 0 and ends the process, exit 0"""
    import subprocess
    script = _script(tmp_path, f"""
        import sys, time, json
        sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
        import bench

        class Eng:
            lib_comm_log = []
            def enable_library_comm(self, timeout, cross_check):
                time.sleep(600)
        out = {{"metric": "PDHG iterations/sec", "value": 123.0, "config": {{"workload": "w"}}, "timing": {{}}}}
        bench.library_phase(out, {{"path": "torch.distributed loop"}}, Eng(), 20, None, 0.2, rank=int(sys.argv[1]), limit=0.3, slack=0.4)
        print("not reached")
    """)
    for rank, want_line in ((0, True), (1, False)):
        r = subprocess.run([sys.executable, script, str(rank)], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and "not reached" not in r.stdout and "hung" in r.stderr
        if want_line:
            d = json.loads(r.stdout.strip())
            assert d["value"] == 123.0 and "abandoned by the watchdog" in d["config"]["exchange"]["path"]
        else:
            assert r.stdout.strip() == ""


# ---------------------------------------------------------------------------------------------------
# the direct-exchange phase (HIP IPC between the ranks): adopted only if connected, cross-checked and faster; both of its forms are
# timed; whatever happens, the line measured on the loop survives
# ---------------------------------------------------------------------------------------------------
class _PeerEng:
    peer_log = ["peer log"]
    peer_local_first = False

    def __init__(self, behaviour):
        self.behaviour, self.calls = behaviour, []

    def enable_peer_exchange(self, cross_check, timeout_ms):
        assert cross_check and timeout_ms > 0
        if self.behaviour == "raise":
            raise RuntimeError("boom")
        return self.behaviour == "on"

    def _peer_check(self):
        self.calls.append("check")

    PEER_FORMS = ("whole product after the wait", "own-block panels between signal and wait", "push beside the own-block panels")
    peer_form = 0

    def set_peer_form(self, form):
        self.peer_form = form
        self.calls.append(("form", form))

    def set_peer_exchange(self, on):
        self.calls.append(("exchange", bool(on)))


def _regions(*norms):
    it = iter(norms)
    return lambda: dict(norm_elapsed=next(it), elapsed=0.1, checks_in=0, restarts_in=0, check_s=0.001)


def test_peer_phase_outcomes():
    # (behaviour, timed regions: whole-product form / own-block-first form / push form, adopted?, value, form kept)
    for behaviour, norms, adopted, value, form in (("off", (), False, 100.0, 0), ("raise", (), False, 100.0, 0),
                                                   ("on", (0.4,), False, 100.0, 0),                  # slower than the loop: not adopted
                                                   ("on", (0.1, 0.2, 0.3), True, 200.0, 0),          # faster; the other forms are slower
                                                   ("on", (0.1, 0.05, 0.08), True, 400.0, 1),        # own-block-first wins
                                                   ("on", (0.1, 0.09, 0.04), True, 500.0, 2)):       # the push form wins
        out, ex, eng = _line(), {"path": "torch.distributed loop"}, _PeerEng(behaviour)
        got = bench.peer_phase(out, ex, eng, 20, _regions(*norms), 0.2, rank=0, first_region_s=0.0)
        assert got is adopted and out["value"] == value, (behaviour, norms, ex)
        if adopted:
            assert "direct exchange" in ex["path"] and ex["torch_loop_value"] == 100.0 and ex["direct_log"] == ["peer log"]
            assert ex["direct_exchange_value"] == round(20 / norms[0], 3) and ex["direct_exchange_value_own_block_first"] == round(20 / norms[1], 3)
            assert ex["direct_exchange_value_push"] == round(20 / norms[2], 3)
            assert eng.peer_form == form and _PeerEng.PEER_FORMS[form] in ex["path"]
        else:
            assert ex["path"] == "torch.distributed loop" and "direct_exchange" in ex
            if behaviour == "on":
                assert ("exchange", False) in eng.calls          # the slower driver is switched off again on every rank


def test_a_hung_peer_phase_still_prints_the_measured_line(tmp_path):
    """a rank stuck inside hipIpcOpenMemHandle (ROCm 7.2 does that for some allocation sizes): the watchdog prints the loop's line"""
    import subprocess
    script = _script(tmp_path, f"""
        import sys, time, json
        sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
        import bench

        class Eng:
            peer_log = []
            def enable_peer_exchange(self, cross_check, timeout_ms):
                time.sleep(600)
        out = {{"metric": "PDHG iterations/sec", "value": 123.0, "config": {{"workload": "w"}}, "timing": {{}}}}
        bench.peer_phase(out, {{"path": "torch.distributed loop"}}, Eng(), 20, None, 0.2, rank=int(sys.argv[1]), first_region_s=0.0)
        print("not reached")
    """)
    env = dict(os.environ, PDLP_PEER_PHASE_DEADLINE="0.5")
    for rank, want_line in ((0, True), (1, False)):
        r = subprocess.run([sys.executable, script, str(rank)], capture_output=True, text=True, timeout=60, env=env)
        assert r.returncode == 0 and "not reached" not in r.stdout and "hung" in r.stderr
        if want_line:
            d = json.loads(r.stdout.strip())
            assert d["value"] == 123.0 and "abandoned by the watchdog" in d["config"]["exchange"]["path"]
        else:
            assert r.stdout.strip() == ""


def test_deadman_can_be_disarmed(capfd):
    import time
    dm = bench.Deadman(0)
    dm.arm(0.2, lambda: "never printed", "test")
    dm.disarm()
    time.sleep(0.5)
    out, _ = capfd.readouterr()
    assert out == ""
