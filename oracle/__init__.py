"""CPU oracle for the SimANS/co_training bi-encoder hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``simxns_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker.

Parity pin: the reference (microsoft/SimXNS) ships no tests and no golden
vectors; its arithmetic lives in third-party HuggingFace ``transformers`` /
``torch``.  The oracle is therefore pinned against *outputs of the reference
itself run in the build container* (``oracle/make_golden.py`` imports
``/root/reference/SimANS/model/models.py`` etc. and commits the vectors under
``tests/golden/``) -- see DESIGN.md section 7 row (c).
"""
