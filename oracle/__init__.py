"""CPU oracle for the ConvONet-Opt restoration path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
product path: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.
The product (``if-defense_amd``) never imports this package and fails loudly
when its HIP library is missing.

Pinning status: the reference (Wuziyi616/IF-Defense) ships no tests, golden
vectors or fixtures for this path, so the oracle is pinned
against outputs of the *reference's own Python modules run in the build
container* (``tests/golden/make_golden.py`` imports them from /root/reference and
writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` replays them).
The trained checkpoint is a Google-Drive download that is not available
offline, so those fixtures use seeded random weights of the same architecture.
"""
