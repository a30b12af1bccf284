"""CPU oracle for the protein-transformer training hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

It is a CPU restatement (plain PyTorch-CPU fp32 ops, one protein at a time,
one atom at a time) of the reference algorithm for the path named by
BASELINE.json `north_star`:

    encoder-only Transformer -> atan2 -> NeRF all-atom build -> dRMSD
    -> backward -> clip -> SGD/Adam step

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it, and only as the checker / the timed CPU baseline.
The product package `protein_transformer_amd` never imports it and has no
CPU fallback: it raises if the HIP extension is missing.

Parity pinning: every function here is checked against golden vectors captured
by importing the upstream reference in the build container
(`tests/golden/make_golden.py` -> `tests/golden/*.npz`, test
`tests/test_oracle_golden.py`).  The eval-only `rmsd` (ProDy Kabsch) is the one
exception: ProDy is not installed anywhere we can run, so `kabsch_rmsd` is
"parity unpinned" and says so.
"""
