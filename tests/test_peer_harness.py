"""The multi-process GPU parity harness (tests/test_gpu_peers.py:run_group) must not be able to hide a mismatch:
torchrun SIGTERMs every other rank the moment one rank exits, so "one rank reported an AssertionError, the rest
produced no result" is what a parity failure looks like -- it fails the test, it is never retried.  (Round 3 retried
exactly that case three times and reported green.)  CPU test of the classification."""
import pytest

from tests.test_gpu_peers import _classify, _StartupFlake

OK = lambda r: {"rank": r, "ok": True, "checks": 3, "runs": 1}      # noqa: E731


def test_all_ranks_fine():
    _classify(3, [OK(0), OK(1), OK(2)], "", "")


def test_one_rank_reports_a_mismatch_and_the_rest_were_torn_down():
    bad = {"rank": 4, "ok": False, "error": "AssertionError(\"join_upsize_3_to_5 rank 4 event 97 ('QUIESCE',) replica 4: apply_count 50 vs 0\")"}
    with pytest.raises(AssertionError, match="apply_count 50 vs 0"):
        _classify(5, [bad], "", "Sending process 2287 closing signal SIGTERM")


def test_a_mismatch_beside_connection_errors_of_the_ranks_it_took_down():
    bad = {"rank": 1, "ok": False, "error": "AssertionError('replica 1: offsets differ')"}
    torn = {"rank": 0, "ok": False, "error": "RuntimeError('Connection closed by peer [127.0.0.1]:4242')"}
    with pytest.raises(AssertionError, match="offsets differ"):
        _classify(3, [bad, torn, OK(2)], "", "")


def test_engine_errors_are_failures_too():
    with pytest.raises(AssertionError, match="EngineError"):
        _classify(3, [{"rank": 2, "ok": False, "error": "EngineError('rep_start rc=-4')"}, OK(0), OK(1)], "", "")


def test_only_the_gloo_start_up_signature_is_retried():
    torn = {"rank": 0, "ok": False, "error": "RuntimeError('Connection closed by peer [127.0.0.1]:4242')"}
    with pytest.raises(_StartupFlake):
        _classify(3, [torn, OK(1), OK(2)], "", "")
    with pytest.raises(_StartupFlake):
        _classify(3, [OK(1), OK(2)], "", "[c10d] Gloo connectFullMesh failed with ...")


def test_ranks_lost_without_a_word_fail():
    with pytest.raises(AssertionError, match="nobody said why"):
        _classify(3, [OK(1), OK(2)], "", "Signal 11 (SIGSEGV) received by PID 77")
