#!/usr/bin/env python3
"""bench.py with a FAULTY Pipeline, for tests/test_gpu_bench_flow.py only (the hooks used to sit inside the timed Pipeline.step):
  VQ_TEST_FAULT=drop_post_wait     the composite's stream does NOT wait for the post kernel of its frame — against an asynchronous transport the frame comes out wrong
  VQ_TEST_FAULT=overlap_timeout    the watchdog's probe reports a timeout on a healthy box — the fall-back to one stream order is exercised
Same command line as bench.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

FAULT = os.environ.get("VQ_TEST_FAULT", "")


class FaultyPipeline(bench.Pipeline):
    def order_composite_behind_post(self, s_comp, b):
        if FAULT != "drop_post_wait":
            super().order_composite_behind_post(s_comp, b)

    def completes_within(self, n_steps, seconds):
        ok = super().completes_within(n_steps, seconds)
        return ok and FAULT != "overlap_timeout"


if __name__ == "__main__":
    assert FAULT in ("drop_post_wait", "overlap_timeout"), "VQ_TEST_FAULT must name the fault"
    bench.Pipeline = FaultyPipeline
    bench.main()
