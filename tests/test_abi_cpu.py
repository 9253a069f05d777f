"""CPU: the C-ABI library builds/loads and exports every symbol include/vdr.h declares; without a
GPU every compute entry point fails loudly (no CPU fallback exists behind the boundary)."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "vdr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vdr_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_declare_the_same_symbols():
    from vdr import _lib
    assert _declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    from vdr import _lib
    lib = _lib.load()  # raises if the .so or any symbol is missing
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.vdr_abi_version() == 8
    assert lib.vdr_kernel_class_name(4) == b"attention"


def test_config_struct_layout_matches_header():
    from vdr import _lib
    assert C.sizeof(_lib.vdr_config) == 4 * 25  # 24 int32 fields + one float


def test_invalid_configs_are_rejected_before_touching_a_device():
    import vdr
    from vdr import _lib
    lib = _lib.load()
    h = C.c_void_p()
    bad = vdr.VdrConfig(dim=100, heads=2)  # head dim != 64
    cc = bad.to_c()
    assert lib.vdr_create(C.byref(cc), 0, C.byref(h)) == -7
    assert b"head dim" in lib.vdr_last_error(None)
    cc = vdr.VdrConfig(img=225, patch=16).to_c()
    assert lib.vdr_create(C.byref(cc), 0, C.byref(h)) == -1
    assert lib.vdr_create(None, 0, C.byref(h)) == -1


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a box without a GPU")
def test_no_gpu_means_loud_failure_not_fallback():
    import vdr
    from vdr import _lib
    lib = _lib.load()
    assert lib.vdr_device_count() == 0
    h = C.c_void_p()
    cc = vdr.VdrConfig().to_c()
    assert lib.vdr_create(C.byref(cc), 0, C.byref(h)) == -2  # VDR_ERR_NO_DEVICE
    assert b"no CPU path" in lib.vdr_last_error(None)
    with pytest.raises(vdr.VdrError):
        vdr.Engine(vdr.VdrConfig())
    buf = (C.c_char * 64)()
    assert lib.vdr_op_attention(buf, buf, 1, 1, 1, 0, None) == -2
    assert lib.vdr_op_layernorm(buf, 0, buf, 0, buf, buf, 1, 4, 1e-5, None) == -2


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "vit-deep-radiomics_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.replace("oracle's", ""), f"{f} mentions the oracle"


def test_no_wide_buffer_store_is_followed_by_a_write_of_its_data():
    """gfx950 hazard found in round 3 (tools/hazard_scan.py, tools/micro/gemm_ring4d_experiment.hip): a buffer_store_dwordx4
    with an SGPR soffset followed at once by a VALU write of one of its data registers can store the new value, and hipcc
    inserts no wait state for that form.  The kernels that use such stores (everything that includes gemm_epi.h) are
    compiled to ISA here and scanned: none may contain the pattern."""
    import glob
    import importlib.util
    spec = importlib.util.spec_from_file_location("hazard_scan", os.path.join(ROOT, "tools", "hazard_scan.py"))
    hs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hs)
    srcs = [f for f in sorted(glob.glob(os.path.join(hs.CSRC, "*.hip")))
            if "raw_buffer_store" in open(f).read() or "gemm_kernels.h" in open(f).read()]
    assert srcs
    total = 0
    for s in hs.compile_isa(srcs, jobs=4):
        n, hits = hs.scan(s, 4)
        total += n
        assert not hits, hits[:3]
    assert total > 0  # the scan saw the stores it is about


def test_no_asm_memory_instruction_reads_an_sgpr_a_valu_has_just_written():
    """The cause of round 3's ring4d memory-access faults (tools/micro/gemm_ring4d_experiment.hip, header): hipcc inserts no
    hazard wait states INSIDE an asm statement, and `VALU writes SGPR -> vector-memory instruction reads it` needs 5 -- a
    pointer reloaded from an SGPR spill (v_readlane) right ahead of an opaque atomic gave it a stale base address.  Every
    kernel of the library that issues memory instructions from asm statements (the opaque LDS-DMA of the attention kernels)
    is compiled to ISA and scanned for that pattern."""
    import glob
    import importlib.util
    spec = importlib.util.spec_from_file_location("hazard_scan", os.path.join(ROOT, "tools", "hazard_scan.py"))
    hs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hs)
    srcs = [f for f in sorted(glob.glob(os.path.join(hs.CSRC, "*.hip"))) if "glds16_raw" in open(f).read() or "asm volatile(\"global_" in open(f).read()]
    assert srcs
    total = 0
    for s in hs.compile_isa(srcs, jobs=4):
        n, hits = hs.scan_asm_sgpr(s)
        total += n
        assert not hits, hits[:3]
    assert total > 0


def test_8phase_kernels_have_no_vector_memory_instruction_their_counted_waits_do_not_know():
    """csrc/gemm_8p.hip waits with COUNTED `s_waitcnt vmcnt(n)` for the LDS-DMA piece a phase needs -- loads, LDS-DMA pieces and
    stores retire one counter in issue order, so every count in the source is exact only while the kernel issues no vector-memory
    instruction the source does not show: no scratch (a spill reload is a load), no plain global load.  The four instantiations
    are compiled to ISA: scratch size 0, and every vector-memory instruction is an LDS-DMA, a buffer store or (none today) an
    atomic."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("hazard_scan", os.path.join(ROOT, "tools", "hazard_scan.py"))
    hs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hs)
    (isa,) = hs.compile_isa([os.path.join(hs.CSRC, "gemm_8p.hip")], jobs=1)
    txt = open(isa).read()
    kernels = re.findall(r"^(_ZN3vdr\S*gemm_8p_kernel\S*):\s*; @\S+\n(.*?); ScratchSize: (\d+)", txt, re.S | re.M)
    assert len(kernels) == 4, [k[0] for k in kernels]
    for name, body, scratch in kernels:
        assert int(scratch) == 0, (name, scratch)
        vmem = re.findall(r"^\s+((?:global|buffer|flat|scratch)_\w+)", body, re.M)
        assert vmem and set(vmem) <= {"global_load_lds_dwordx4", "buffer_store_dwordx4"}, (name, sorted(set(vmem)))
