#!/usr/bin/env python3
"""Host-side timeline of the test loop under the drop-in brats_test_default script (tools/script_throughput.py) with the reference's batches of 32
slices, no coalescing: when the main thread waits for the loader thread (`next(loader)`), enqueues a batch (`_run_steps`), waits for a batch's
outputs (`wait`) and finishes it (`_finish_batch`: assembly, subject steps, hooks) -- the entries of half a second in the steady state that
took more than 4 ms.  This is what showed that one batch of run-ahead left the GPU idle while the host finished a subject (round 4).

    python tools/loop_timeline.py [subjects, default 8]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
os.environ['RCU_SCRIPT_PROFILE'] = '0'
from rcu_amd import loops
log = []
T0 = time.perf_counter()
def wrap(cls, name):
    inner = getattr(cls, name)
    def f(self, *a, **k):
        t = time.perf_counter()
        r = inner(self, *a, **k)
        log.append((name, t - T0, time.perf_counter() - t))
        return r
    setattr(cls, name, f)
wrap(loops.Test, '_run_steps'); wrap(loops.Test, '_finish_batch'); wrap(loops._Download, 'wait')
_pf = loops.prefetch
def timed_prefetch(*a, **k):
    k['timing'] = True      # the loader thread logs where its time went
    it = _pf(*a, **k)
    while True:
        t = time.perf_counter()
        try:
            item = next(it)
        except StopIteration:
            return
        log.append(('next(loader)', t - T0, time.perf_counter() - t))
        yield item
loops.prefetch = timed_prefetch
import script_throughput
sys.argv = ['x', sys.argv[1] if len(sys.argv) > 1 else '8', '20', '32', '0']
script_throughput.main()
ev = [e for e in log]
start = [e for e in ev if e[0] == '_run_steps']
base = start[15][1]
for name, t, d in ev:
    if base <= t <= base + 0.6 and d > 0.004:
        print('%-14s at %7.1f ms  took %6.1f ms' % (name, (t - base) * 1e3, d * 1e3))
