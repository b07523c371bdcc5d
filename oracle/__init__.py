"""CPU oracle for the MC-dropout / ensemble uncertainty path.  TEST INFRASTRUCTURE ONLY.

A plain restatement (torch-CPU functional ops + numpy, plus a small C file for the integer
histograms) of what the reference computes on this path; every function cites the reference
file:line it follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package, and there only as the checker / the reported CPU
baseline -- never as the product.  The product (``reliability-challenges-uncertainty_amd``) does
not import it and fails loudly when its HIP library is missing.

Pinning: the reference has no tests of its own, so the oracle is pinned
against golden vectors produced by importing the reference in the build container
(``tests/golden/generate_golden.py`` -> ``tests/golden/*.npz``, fixtures G1..G11 of SURVEY.md
section 8c); ``tests/test_oracle_golden.py`` checks every one of them.  Parts with no executable
definition in the reference tree (pymia's ConfusionMatrix/Dice/Accuracy arithmetic) are restated
from the call sites and marked "parity unpinned" where they are defined.
"""
