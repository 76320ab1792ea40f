"""Experiment (run once on the GPU box): does HIP police another thread's hipEventQuery during a stream capture, and does
capture_error_mode="thread_local" lift that?  This is the question behind training/ddp.py's former `time.sleep(0.3)`:
c10d's watchdog thread polls the events of outstanding collectives with hipEventQuery about every 100 ms.

    python scratch/capture_mode_probe.py            # runs each mode in a child process (a failed capture may poison a context)
    python scratch/capture_mode_probe.py global     # one mode, in this process

Prints one JSON line per mode: what the polling thread saw (query results or the error) and whether the capture survived."""
import json
import subprocess
import sys
import threading
import time


def run(mode: str) -> dict:
    import torch
    x = torch.zeros(1 << 20, device="cuda")
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    res = {"mode": mode, "queries": 0}
    start, done = threading.Event(), threading.Event()

    def poll():  # the watchdog's role: a foreign thread (default capture mode of ITS thread: global) querying a finished event
        start.wait()
        try:
            for _ in range(40):
                ev.query()
                res["queries"] += 1
                time.sleep(0.002)
        except Exception as e:  # noqa: BLE001
            res["poll_error"] = repr(e)[:300]
        done.set()

    t = threading.Thread(target=poll)
    t.start()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=s, capture_error_mode=mode):
            start.set()
            for _ in range(20):
                x.add_(1)
            done.wait()
        g.replay()
        torch.cuda.synchronize()
        res["capture"] = "ok"
        res["x0"] = float(x[0])
    except Exception as e:  # noqa: BLE001
        res["capture"] = repr(e)[:300]
        start.set()
    t.join()
    return res


if __name__ == "__main__":
    if len(sys.argv) > 1:
        print(json.dumps(run(sys.argv[1])))
    else:
        for mode in ("thread_local", "relaxed", "global"):
            p = subprocess.run([sys.executable, __file__, mode], capture_output=True, text=True, timeout=300)
            out = [l for l in p.stdout.splitlines() if l.startswith("{")]
            print(out[-1] if out else json.dumps({"mode": mode, "rc": p.returncode, "stderr": p.stderr[-400:]}))
