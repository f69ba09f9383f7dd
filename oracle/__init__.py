"""CPU oracle for the MSML hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker or as the timed CPU
baseline -- never as the thing shipped.  The product path (``msml_amd``)
fails loudly when its HIP library is missing and never routes through here.

The oracle is a plain torch-CPU fp32 restatement of the reference's
arithmetic, written from the formulas in SURVEY.md Appendix A with the same
module / parameter names as the reference, so that a state dict produced by
either side loads ``strict=True`` into the other.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference
(container only, read-only) and records golden vectors under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement
against every one of them.  The reference itself has no tests.
"""
