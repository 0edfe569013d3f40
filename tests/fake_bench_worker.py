"""Stand-in for bench.py's worker in the CPU tests of its supervisor (tests/test_bench_cpu.py): behaves as FAKE_MODE says."""
import json
import os
import sys
import time

mode = os.environ["FAKE_MODE"]
attempt = int(os.environ["NBMF_BENCH_ATTEMPT"])
assert os.environ["NBMF_BENCH_WORKER"] == "1" and os.environ["NBMF_RDZV_GENERATION"] == str(attempt)


def note(text):
    with open(os.environ["NBMF_BENCH_NOTE"], "w") as f:
        f.write(text)


if mode == "ok" or attempt == 1 and mode in ("fail_then_ok", "hang_then_ok", "crash_then_ok"):
    print(json.dumps({"attempt": attempt, "argv": sys.argv[1:], "failed": os.environ.get("NBMF_BENCH_FAILED")}), flush=True)
    sys.exit(0)
if mode == "fail_then_ok":
    note("NBMFHipError: peer exchange timed out")
    sys.exit(75)
if mode == "hang_then_ok":
    time.sleep(600)
if mode == "crash_then_ok":
    os.kill(os.getpid(), 11)
if mode == "always_fail":
    note(f"attempt {attempt} broke")
    sys.exit(3)
raise SystemExit(f"unknown FAKE_MODE {mode}")
