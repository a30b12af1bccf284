"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol the header
declares, the ctypes table covers the header, and the product never falls back to a CPU path."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ptamd.h")
PKG = os.path.join(ROOT, "protein_transformer_amd")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptamd_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built_lib():
    from protein_transformer_amd import _lib, build
    build.build()                       # hipcc cross-compiles gfx950 without a GPU
    return _lib


def test_library_exports_every_declared_symbol(built_lib):
    handle = ctypes.CDLL(built_lib.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 30
    for name in names:
        assert hasattr(handle, name), f"{name} is declared in include/ptamd.h but not exported by libptamd.so"


def test_ctypes_table_matches_header(built_lib):
    built_lib.lib()
    assert not built_lib.MISSING
    assert sorted(built_lib.SIGNATURES) == declared_functions()


def test_no_cuda_symbols_or_hipify_shims():
    out = subprocess.run(["grep", "-rIl", "-E", "cuda_runtime|__HIP_PLATFORM|hipify|triton", os.path.join(PKG, "csrc")],
                         capture_output=True, text=True).stdout.strip()
    assert out == "", out


def test_host_only_entry_points(built_lib):
    lib = built_lib.lib()
    assert lib.ptamd_version().startswith(b"ptamd")
    assert [lib.ptamd_sidechain_atoms(r) for r in range(20)] == [1, 2, 4, 5, 7, 0, 6, 4, 5, 4, 4, 4, 3, 5, 7, 2, 3, 3, 10, 8]
    assert lib.ptamd_sidechain_atoms(20) == -1 and lib.ptamd_sidechain_atoms(-1) == -1
    assert lib.ptamd_nerf_workspace_bytes(32, 512) == 32 * 512 * 12 * 4
    assert lib.ptamd_drmsd_workspace_bytes(32, 512) > 32 * 512 * 14 * 52
    assert lib.ptamd_gemm_workspace_bytes(512, 512, 1) == (512 + 512) * 4    # row scales of the f16x2 arithmetic
    assert lib.ptamd_gemm_workspace_bytes(512, 512, 8) == (8 * 512 * 512 + 8 * 16 * 512) * 4 + (512 + 512) * 4   # + C / column-sum slabs
    assert lib.ptamd_gemm_workspace_bytes(510, 30, 1) == (512 + 32) * 4
    # argument validation happens on the host, before any launch
    assert lib.ptamd_nerf_fwd(None, None, 0, 5, None, None, None) == -1          # PTAMD_ERR_BAD_SHAPE
    assert lib.ptamd_nerf_fwd(None, None, 2, 5000, None, None, None) == -2       # PTAMD_ERR_TOO_LONG
    assert lib.ptamd_drmsd_fwd_bwd(None, None, None, 2, 8, None, None, None, 0, None) == -3   # PTAMD_ERR_WORKSPACE
    assert lib.ptamd_attention_fwd(None, None, 1, 8, 2, 24, 0.0, 0, 0, 4, None, None, None, None, None, None) == -1      # head size 24
    assert lib.ptamd_attention_fwd(None, None, 1, 8, 2, 32, 0.0, 0, 0, 7, None, None, None, None, None, None) == -1      # unknown arithmetic
    assert lib.ptamd_kabsch_rmsd(None, None, None, 0, 8, None, None) == -1


def test_gemm_arithmetic_policy(built_lib):
    """`arith` of ptamd_gemm_args / ptamd_gemm_products (host only): which arithmetic a call runs in (include/ptamd.h).
    The library has no process-wide mode: there is nothing to set, the answer depends on the arguments alone."""
    import ctypes as C
    from protein_transformer_amd import kernels as K
    lib = built_lib.lib()
    assert not hasattr(lib, "ptamd_gemm_set_mode") and not hasattr(lib, "ptamd_gemm_get_mode")

    def products(arith, a_kmajor, Kd=512, lda=512):
        args = built_lib.GemmArgs(M=256, N=128, K=Kd, A=None, lda=lda, a_kmajor=a_kmajor, B=None, ldb=Kd, b_kmajor=0,
                                  C=None, ldc=128, arith=arith)
        return lib.ptamd_gemm_products(C.byref(args))
    for mode, want in [(K.GEMM_F32, (1, 1)), (K.GEMM_BF16X3, (6, 6)), (K.GEMM_BF16X3_FULL, (9, 9)), (K.GEMM_F16X2, (3, 3)),
                       (K.GEMM_AUTO, (3, 6))]:                   # (K-contiguous A, k-major A)
        assert (products(mode, 0), products(mode, 1, lda=256)) == want, mode
        assert products(mode, 0, Kd=8, lda=8) == 1               # K < 16 always runs on the exact-f32 MFMA
    assert products(99, 0) == -1 and products(-1, 0) == -1       # PTAMD_ERR_BAD_SHAPE: unknown arithmetic
    assert lib.ptamd_gemm_products(None) == -1
    # the host-side default (kernels.set_gemm_mode) is plain Python state that calls without `arith=` pick up
    old = K.get_gemm_mode()
    try:
        K.set_gemm_mode(K.GEMM_BF16X3)
        assert K.get_gemm_mode() == K.GEMM_BF16X3
        with pytest.raises(ValueError):
            K.set_gemm_mode(99)
        assert K.get_gemm_mode() == K.GEMM_BF16X3
    finally:
        K.set_gemm_mode(old)


def test_library_keeps_no_mutable_state():
    """'re-entrant, no hidden global state'.  No function-static or namespace-scope mutable
    variable in csrc/ besides the thread-local last-HIP-error word."""
    import glob
    import re
    bad = []
    for path in glob.glob(os.path.join(PKG, "csrc", "*.h*")) + glob.glob(os.path.join(PKG, "csrc", "*.cpp")):
        for n, line in enumerate(open(path), 1):
            code = line.split("//")[0]
            if re.match(r"\s+static\s+(?!const|constexpr|inline|__device__|__global__|__forceinline__)", code):
                bad.append(f"{os.path.basename(path)}:{n}: {line.strip()}")
            if re.match(r"^(int|bool|float|double|unsigned|size_t)\s+g_\w+", code):
                bad.append(f"{os.path.basename(path)}:{n}: {line.strip()}")
    assert not bad, bad


def test_product_has_no_cpu_fallback(built_lib):
    from protein_transformer_amd import losses
    from protein_transformer_amd.models.encoder_only import EncoderOnlyTransformer
    from protein_transformer_amd.protein.Sequence import VOCAB
    with pytest.raises(RuntimeError, match="device tensors only"):
        losses.inverse_trig_transform(torch.zeros(1, 3, 24))
    with pytest.raises(RuntimeError, match="device tensors only"):
        losses.drmsd_forward_backward(torch.zeros(1, 28, 3), torch.zeros(1, 28, 3), torch.zeros(1, 2, dtype=torch.int64))
    m = EncoderOnlyTransformer(1, 4, 32, 64, 16, VOCAB, [0.1] * 24, True)
    with pytest.raises(RuntimeError, match="MI355X only"):
        m(torch.zeros(1, 8, dtype=torch.int64))


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from protein_transformer_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="is missing"):
        _lib.lib()


def test_product_never_imports_the_oracle():
    hits = subprocess.run(["grep", "-rIn", "-E", r"^\s*(from|import)\s+oracle", PKG], capture_output=True, text=True).stdout
    assert hits.strip() == "", hits


def integration_stub_source():
    """The first python block of INTEGRATION.md: the reference-side binding, as printed."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = [b for b in blocks if "def batch_drmsd_and_grad" in b]
    assert len(stub) == 1
    return stub[0]


def test_integration_md_stub_matches_header(built_lib):
    """The ctypes `argtypes` INTEGRATION.md prints for the reference-side stub, against include/ptamd.h: same entry points,
    same number and kinds of arguments (pointer / int / int64 / size_t) - without a GPU (the stub is EXECUTED on one by
    tests/test_gpu_loss_path.py::test_integration_md_stub_runs)."""
    src = integration_stub_source()
    header = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    kinds = {"_p": "ptr", "_i": "int", "_sz": "size_t", "ctypes.c_int64": "int64"}
    seen = 0
    for name, args in re.findall(r"_lib\.(ptamd_[a-z0-9_]+)\.argtypes = \[(.*?)\]", src):
        got = [kinds[a.strip()] for a in args.split(",")]
        decl = re.search(r"\b" + name + r"\s*\((.*?)\)\s*;", header, flags=re.S)
        assert decl, name
        want = []
        for a in decl.group(1).split(","):
            a = " ".join(a.split())
            want.append("ptr" if "*" in a else "int64" if a.startswith("int64_t") else "size_t" if a.startswith("size_t") else "int")
        assert got == want, (name, got, want)
        seen += 1
    assert seen == 5
    for name in re.findall(r"_lib\.(ptamd_[a-z0-9_]+)", src):
        assert name in built_lib.SIGNATURES, name
