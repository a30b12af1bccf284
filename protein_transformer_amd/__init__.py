"""protein_transformer_amd: the protein-transformer training hot path on MI355X (gfx950).

Module layout mirrors the reference package (`losses`, `dataset`, `train`,
`protein.Structure`, `protein.Sequence`, `models.encoder_only`) so that it is a drop-in
for that path; the arithmetic lives in hand-written HIP kernels (`csrc/`) behind the
C ABI of `include/ptamd.h`.
"""
__version__ = "0.1.0"
