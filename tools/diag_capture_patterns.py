"""Which stream patterns inside a hipGraph capture does this HIP runtime survive?  Each pattern runs in its own process
(a crash is a segfault in hipStreamEndCapture): python tools/diag_capture_patterns.py [A|B|C|D]; no argument = all, as children."""
import subprocess
import sys

import torch


def run(p):
    x = torch.zeros(1024, device="cuda")
    s = torch.cuda.Stream()
    s2 = torch.cuda.Stream()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        m = torch.cuda.current_stream()
        x.add_(1)
        if p == "A":    # fork S, join, fork the SAME stream again, join
            s.wait_stream(m)
            with torch.cuda.stream(s):
                y = x * 2
            m.wait_stream(s)
            x.add_(1)
            s.wait_stream(m)
            with torch.cuda.stream(s):
                z = x * 3
            m.wait_stream(s)
        elif p == "B":  # fork S; S waits on M AGAIN while it is already part of the capture; join
            s.wait_stream(m)
            with torch.cuda.stream(s):
                y = x * 2
            x.add_(1)
            s.wait_stream(m)
            with torch.cuda.stream(s):
                z = x * 3 + y
            m.wait_stream(s)
        elif p == "C":  # two different side streams, each forked once and joined once (round 4's pattern, twice)
            s.wait_stream(m)
            with torch.cuda.stream(s):
                y = x * 2
            x.add_(1)
            s2.wait_stream(m)
            with torch.cuda.stream(s2):
                z = x * 3
            m.wait_stream(s)
            m.wait_stream(s2)
        elif p == "D":  # a side stream with TWO incoming edges at its start (waits on M and on the other side stream)
            s.wait_stream(m)
            with torch.cuda.stream(s):
                y = x * 2
            x.add_(1)
            s2.wait_stream(m)
            s2.wait_stream(s)
            with torch.cuda.stream(s2):
                z = x * 3 + y
            m.wait_stream(s2)
        elif p in ("E", "F"):  # round 5's sequence: fork S, join, fork S again, S waits on M again, join (F: the second half on another thread)
            s.wait_stream(m)
            with torch.cuda.stream(s):
                y = x * 2
            ev = torch.cuda.Event()
            ev.record(s)

            def second_half():
                with torch.cuda.stream(m):
                    m.wait_event(ev)
                    x.add_(1)
                    s.wait_stream(m)
                    with torch.cuda.stream(s):
                        z = x * 3
                    x.add_(1)
                    ev2 = torch.cuda.Event()
                    ev2.record(m)
                    s.wait_event(ev2)
                    with torch.cuda.stream(s):
                        w = x * 4 + z
                    x.add_(1)
                    m.wait_stream(s)

            if p == "E":
                second_half()
            else:
                import threading

                t = threading.Thread(target=second_half)
                t.start()
                t.join()
        x.add_(1)
    g.replay()
    torch.cuda.synchronize()
    print(p, "ok", float(x[0]), flush=True)


if len(sys.argv) > 1:
    run(sys.argv[1])
else:
    for p in "ABCDEF":
        r = subprocess.run([sys.executable, __file__, p], capture_output=True, text=True)
        print(p, "rc", r.returncode, (r.stdout.strip().splitlines() or ["-"])[-1][:80], flush=True)
