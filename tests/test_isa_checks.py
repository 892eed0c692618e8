"""Checks on the generated gfx950 code that no run-time test can make deterministic (CPU: hipcc cross-compiles)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import check_isa  # noqa: E402


def test_chase_progress_counter_is_published_behind_a_drain():
    """the bulge chase's counter store must not overtake the step's band stores (sb2st.hip, `publish`)."""
    assert len(check_isa.check_chase_publish()) >= 2


def test_diag_kernel_barriers_follow_a_drain_of_the_flag_stores():
    """round 4: the round-4 diagonal-block kernel's role hand-out -- inline-asm flag stores must be drained before the barrier that
    publishes them (the compiler does not do it for inline asm)"""
    assert len(check_isa.check_diag_barriers()) == 2
