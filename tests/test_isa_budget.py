"""Build-quality guard for the N = 4096 transform kernels (no GPU needed: hipcc cross-compiles gfx950 here).

Two properties the throughput depends on and that an innocent-looking edit can lose without any test failing
("ks_last_ntt"):

  * occupancy: the one-transform-per-workgroup kernels must fit 128 VGPRs (4 waves per SIMD) without scratch, the
    upper-level kernel 256 (2 waves per SIMD) without scratch;
  * the product loops (k digits x key) must have the ~33 loads of one digit in flight TOGETHER.  With a few more
    registers live across the loop the compiler falls back to one `s_waitcnt vmcnt(0)` per load -- 33 dependent memory
    round trips -- which once made the products cost more than the transform.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pir_amd", "csrc", "ntt_kernels.hip")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _compile(tmp_path_factory, logn, pack_bytes=5):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / ("ntt%d_p%d.s" % (logn, pack_bytes))
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-DPIRGPU_LOGN=%d" % logn,
                    "-DPIRGPU_PACK_BYTES=%d" % pack_bytes, "--cuda-device-only", "-S", SRC, "-o", str(out)], check=True,
                   capture_output=True, timeout=600)
    return out.read_text().split("\n")


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    return _compile(tmp_path_factory, 12)


@pytest.fixture(scope="module")
def isa13(tmp_path_factory):
    return _compile(tmp_path_factory, 13)


def _function(isa, prefix):
    start = next(i for i, l in enumerate(isa) if l.startswith(prefix) and l.rstrip().split(";")[0].rstrip().endswith(":"))
    end = next(i for i in range(start, len(isa)) if isa[i].startswith(".Lfunc_end"))
    return isa[start:end]


def _descriptor(isa, prefix):
    i = next(i for i, l in enumerate(isa) if ".amdhsa_kernel " + prefix in l)
    block = "\n".join(isa[i:i + 40])
    return (int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", block).group(1)),
            int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", block).group(1)))


P = "_ZN6pirgpu5deg12"
FOUR_WAVE = ["15ks_digit_kernelILi1ELb1ELb0ELb0E", "15ks_digit_kernelILi1ELb1ELb1ELb0E", "18ks_mac_intt_kernelILi1ELb1E",
             "21ks_mac_combine_kernelILi1ELb1ELb0ELb0E", "21ks_mac_combine_kernelILi1ELb1ELb1ELb1E",
             "18ks_last_ntt_kernelILi1ELb1ELb0E", "18ks_last_ntt_kernelILi1ELb1ELb1E", "16ntt_batch_kernelILi1ELb1E",
             "18tree_c0_ntt_kernelILi1ELb1E"]


@pytest.mark.parametrize("kernel", FOUR_WAVE)
def test_transform_kernels_keep_four_waves_per_simd(isa, kernel):
    vgprs, scratch = _descriptor(isa, P + kernel)
    assert scratch == 0, "%s spills %d bytes per lane" % (kernel, scratch)
    assert vgprs <= 128, "%s needs %d VGPRs: 3 waves per SIMD" % (kernel, vgprs)


@pytest.mark.parametrize("kernel", ["18upper_fused_kernelILi1ELb1ELb1E", "18upper_fused_kernelILi1ELb1ELb0E",
                                    "18upper_fused_kernelILi1ELb0ELb0E"])
def test_upper_level_kernel_keeps_two_waves_per_simd(isa, kernel):
    vgprs, scratch = _descriptor(isa, P + kernel)
    assert scratch == 0 and vgprs <= 256, (kernel, vgprs, scratch)


def _loop_load_batches(body):
    """For every loop (label ... backward branch to it): loads issued before the loop's first vmcnt wait."""
    labels = {l.split(":")[0].strip(): i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    batches = []
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loads = 0
            for t in body[labels[m.group(1)]:i]:
                t = t.strip()
                if t.startswith(("global_load", "buffer_load")):
                    loads += 1
                elif t.startswith("s_waitcnt") and "vmcnt" in t:
                    break
            batches.append(loads)
    return batches


@pytest.mark.parametrize("kernel", ["18ks_mac_intt_kernelILi1ELb1E", "21ks_mac_combine_kernelILi1ELb1ELb1ELb1E",
                                    "18ks_last_ntt_kernelILi1ELb1ELb1E", "18ks_last_ntt_kernelILi1ELb1ELb0E"])
def test_product_loop_issues_its_loads_together(isa, kernel):
    batches = _loop_load_batches(_function(isa, P + kernel))
    assert batches and max(batches) >= 30, "%s: loads in flight per loop %s (a digit needs 33 together)" % (kernel, batches)


# N = 8192 (cfg 4: 43/44-bit moduli, digits and tree as doubles)
P13 = "_ZN6pirgpu5deg13"


@pytest.mark.parametrize("kernel", ["18ks_mac_intt_kernelILi1ELb0E", "21ks_mac_combine_kernelILi1ELb0ELb0ELb0E",
                                    "18ks_last_ntt_kernelILi1ELb0ELb0E", "18ks_last_ntt_kernelILi1ELb0ELb1E",
                                    "15ks_digit_kernelILi1ELb0ELb0ELb0E"])
def test_n8192_transform_kernels(isa13, kernel):
    vgprs, scratch = _descriptor(isa13, P13 + kernel)
    assert scratch == 0 and vgprs <= 128, (kernel, vgprs, scratch)
    if "digit" not in kernel:
        batches = _loop_load_batches(_function(isa13, P13 + kernel))
        assert batches and max(batches) >= 30, (kernel, batches)


# N = 16384 (cfg 5: 48/49-bit moduli, wide fp64 flavour): 1024-thread workgroups, ONE per CU, 128 registers per wave
P14 = "_ZN6pirgpu5deg14"


@pytest.fixture(scope="module")
def isa14(tmp_path_factory):
    return _compile(tmp_path_factory, 14)


@pytest.mark.parametrize("kernel,spill", [("15ks_digit_kernelILi2ELb0ELb0ELb0E", 0), ("16upper_ntt_kernelILi2ELb0E", 0),
                                          ("18ks_mac_intt_kernelILi2ELb0E", 0), ("21ks_mac_combine_kernelILi2ELb0ELb0ELb0E", 0),
                                          ("18ks_last_ntt_kernelILi2ELb0ELb0E", 0), ("16ntt_batch_kernelILi2ELb1E", 0),
                                          # the looped forms (several transforms of one source per workgroup) hold the
                                          # source across the transform: upper_ntt fits, ks_digit reloads 4 doubles of
                                          # it per transform from scratch (32 of the 128 KiB it stores)
                                          ("16upper_ntt_kernelILi2ELb1E", 0), ("15ks_digit_kernelILi2ELb0ELb0ELb1E", 48)])
def test_n16384_transform_kernels(isa14, kernel, spill):
    vgprs, scratch = _descriptor(isa14, P14 + kernel)
    assert vgprs <= 128 and scratch <= spill, (kernel, vgprs, scratch)


# ---- the database scan (scan_mfma.hip): register budgets of the two workgroup shapes ---------------------------------
SCAN_SRC = os.path.join(ROOT, "pir_amd", "csrc", "scan_mfma.hip")


@pytest.fixture(scope="module")
def scan_isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "scan.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--cuda-device-only", "-S", SCAN_SRC,
                    "-o", str(out)], check=True, capture_output=True, timeout=600)
    return out.read_text().split("\n")


def _scan_descriptor(isa, L, KS, NW, top4=True, f64_fold=None):
    # the variant a context runs by default: the fp64 fold of the digit diagonals from 6 digits per residue on
    f64_fold = (L >= 6) if f64_fold is None else f64_fold
    name = "_ZN6pirgpu16scan_mfma_kernelILi%dELi%dELi%dELb%dELb%dEEE" % (L, KS, NW, 1 if top4 else 0, 1 if f64_fold else 0)
    i = next(i for i, l in enumerate(isa) if ".amdhsa_kernel " + name in l)
    block = "\n".join(isa[i:i + 40])
    get = lambda key: int(re.search(r"\.amdhsa_%s (\d+)" % key, block).group(1))
    return get("next_free_vgpr"), get("accum_offset"), get("private_segment_fixed_size")


@pytest.mark.parametrize("top4", [True, False])
@pytest.mark.parametrize("L,KS", [(5, 1), (5, 2), (5, 3), (6, 1), (6, 2), (7, 1), (7, 2)])
def test_eight_wave_scan_fits_two_waves_per_simd(scan_isa, L, KS, top4):
    """8-wave workgroups run two waves per SIMD: 256 registers per wave, no scratch (<6, 3, 8> is known to spill two
    registers and is only used when the 4-wave kernel is forced off).  top4: the top digit stored as nibbles (L <= 6)."""
    if top4 and L == 7:
        pytest.skip("the nibble form is not built for L = 7")
    total, _, scratch = _scan_descriptor(scan_isa, L, KS, 8, top4)
    assert total <= 256 and scratch == 0, (L, KS, top4, total, scratch)


@pytest.mark.parametrize("L,KS", [(5, 7), (6, 4), (6, 7), (7, 5)])
def test_four_wave_scan_uses_the_unified_register_file_without_scratch(scan_isa, L, KS):
    """4-wave workgroups run one wave per SIMD and may take the whole 512-entry VGPR + AGPR file: the selectors of up to
    7 k-steps live there (cfg 4: <6, 7, 4>).  More than 256 registers in use shows the AGPR half is really used; no
    scratch (<7, 6, 4>, cfg 5, is the one known exception: 20 spilled registers, still faster than two chunks)."""
    for top4 in (True, False) if L <= 6 else (False,):
        total, accum_offset, scratch = _scan_descriptor(scan_isa, L, KS, 4, top4)
        # the nibble form of <6, 7, 4> (cfg 4) spills 13 registers to unpack into and is still 10 % faster than the
        # byte form (4.6 against 5.1 ms: 8 % fewer bytes to stream)
        assert scratch <= (64 if (L, KS, top4) == (6, 7, True) else 0), (L, KS, top4, scratch)
        assert total <= 512, (L, KS, top4, total)
    total, accum_offset, scratch = _scan_descriptor(scan_isa, L, KS, 4, False)
    assert total <= 512 and (KS < 5 or total > 256), (L, KS, total, accum_offset)


# ---- round 6: the packed intermediates at 6 / 7 bytes per residue (cfg 4: N = 8192, 43 / 44-bit moduli; cfg 5: N = 16384, 48 / 49
# bits) -- the kernels that used to move doubles there now unpack 2 / 3 high bytes per residue and must still fit
P13P6 = "_ZN6pirgpu7deg13p6"
P14P7 = "_ZN6pirgpu7deg14p7"


@pytest.fixture(scope="module")
def isa13p6(tmp_path_factory):
    return _compile(tmp_path_factory, 13, 6)


@pytest.fixture(scope="module")
def isa14p7(tmp_path_factory):
    return _compile(tmp_path_factory, 14, 7)


@pytest.mark.parametrize("kernel", ["18ks_mac_intt_kernelILi1ELb1E", "21ks_mac_combine_kernelILi1ELb1ELb0ELb0ELb0E",
                                    "21ks_mac_combine_kernelILi1ELb1ELb1ELb1ELb0E", "18ks_last_ntt_kernelILi1ELb1ELb1ELi0E",
                                    "15ks_digit_kernelILi1ELb1ELb0ELb0E", "15ks_digit_kernelILi1ELb1ELb1ELb0E"])
def test_n8192_six_byte_kernels(isa13p6, kernel):
    vgprs, scratch = _descriptor(isa13p6, P13P6 + kernel)
    assert scratch == 0 and vgprs <= 128, (kernel, vgprs, scratch)
    if "digit" not in kernel:
        batches = _loop_load_batches(_function(isa13p6, P13P6 + kernel))
        assert batches and max(batches) >= 30, (kernel, batches)


# (the tree stays in doubles at N = 16384: the combine kernel with a packed tree on both sides spills 68 bytes there, ctx.hip)
@pytest.mark.parametrize("kernel", ["18ks_mac_intt_kernelILi2ELb1E", "21ks_mac_combine_kernelILi2ELb1ELb0ELb0ELb0E",
                                    "18ks_last_ntt_kernelILi2ELb1ELb0ELi0E", "15ks_digit_kernelILi2ELb1ELb0ELb0E"])
def test_n16384_seven_byte_kernels(isa14p7, kernel):
    vgprs, scratch = _descriptor(isa14p7, P14P7 + kernel)
    assert vgprs <= 128 and scratch == 0, (kernel, vgprs, scratch)


def test_c0_ntt_kernel_issues_its_product_loads_together(isa):
    """ks_c0_ntt_kernel (option C0_NTT = 2): with the key word loaded inside the product expression the compiler waited for
    every load on its own (33 dependent round trips per digit; the kernel took twice its twin's time, round 6)."""
    batches = _loop_load_batches(_function(isa, P + "16ks_c0_ntt_kernelILi1ELb1ELb1ELb1E"))
    assert batches and max(batches) >= 30, batches
