"""CPU checks of the C-ABI boundary: the library loads without a GPU, exports exactly the symbols declared in
include/ddm_hip.h, fails loudly (no CPU fallback) when no HIP device is present, and the product never imports
the oracle."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "ddm_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ddm_[a-z0-9_]+)\s*\(", txt)) - {"ddm_alltoall_fn", "ddm_allreduce_fn"})


def test_library_exports_every_declared_symbol(ddm):
    lib = ddm.load_library()
    declared = _declared()
    assert len(declared) >= 45
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ddm_hip.h but not exported"
        assert name in ddm.SYMBOLS, f"{name} has no ctypes prototype in the Python binding"
    assert sorted(ddm.SYMBOLS) == declared


def test_no_cpu_fallback_without_device(ddm):
    import torch
    if torch.cuda.is_available():
        return
    lib = ddm.load_library()
    h = ctypes.c_void_p()
    assert lib.ddm_ctx_create(0, None, ctypes.byref(h)) != 0 and not h.value
    try:
        ddm.Context(0)
    except ddm.DdmError:
        pass
    else:
        raise AssertionError("Context() must fail without a HIP device")


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "dune-ddm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hh", ".hpp", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dirpath, f)
                assert "liboracle" not in src


def test_pipe_kernel_has_no_spills():
    """The pipe engine's compute wave keeps gathers in flight behind the compiler's back (inline asm, explicit waits):
    a register spill (scratch or AGPR) of such a register would be silent.  The product builds must have none."""
    import re
    import subprocess
    src = os.path.join(ROOT, "dune-ddm_amd", "csrc", "ddm_hip.hip")
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--cuda-device-only",
                          "-c", "-o", os.devnull, src, "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    blocks = re.split(r"remark: Function Name: ", out.stderr)
    seen = 0
    for b in blocks:
        if not b.startswith("_ZN3ddm11k_trsv_pipe"):
            continue
        name = b.split()[0]
        vals = {k: int(v) for k, v in re.findall(r"(ScratchSize \[bytes/lane\]|AGPRs|VGPRs|VGPRs Spill|SGPRs Spill): (\d+)", b)}
        if name.endswith("k_trsv_pipeILb1EEEvNS_10PipeParamsE"):
            continue                                            # the stamped diagnostic build is not a product path
        seen += 1
        assert vals["ScratchSize [bytes/lane]"] == 0 and vals["AGPRs"] == 0 and vals["VGPRs Spill"] == 0, (name, vals)
        assert vals["VGPRs"] <= 256, (name, vals)              # two workgroups of three waves per CU
    assert seen == 1


def test_bench_scripts_compile_and_keep_the_contract_keys():
    """bench.py / bench_convdiff.py need a GPU to run; here: they compile, and the JSON line they print carries the keys of the driver's
    contract plus the roofline / cpu_baseline objects."""
    import py_compile
    for name in ("bench.py", "bench_convdiff.py"):
        path = os.path.join(ROOT, name)
        py_compile.compile(path, doraise=True)
        src = open(path).read()
        for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"', '"higher_is_better"', '"scaling"', '"vs_baseline"',
                    '"dtype"', '"data"', '"config"', '"roofline"', '"cpu_baseline"', '"bound"', '"achieved"', '"peak"', '"frac"', '"traffic"', '"cores"', '"kind"', '"sample"'):
            assert key in src, (name, key)
