"""The register / scratch budget the measured performance rests on, checked at compile time.

hipcc cross-compiles gfx950 without a GPU and reports every kernel's resources with -Rpass-analysis=kernel-resource-usage.
ANY scratch in the refill trace kernel costs it its occupancy (+8 % when it was removed), and the shade kernel's fourth wave
per SIMD (<= 128 VGPRs) is worth 10 %: a change that silently crosses one of these lines should fail here, not in a benchmark."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "unity_webgpu_pathtracer_amd", "csrc")


def _device_flags(unit_a=False):
    """The flags the product is built with: FLAGS / HIPFLAGS of csrc/Makefile (minus -Wall); unit_a: HIPFLAGS_A, the flags of the
    second compilation of pt_wavefront.hip -- the translation unit the default schedule's trace and shade kernels come from."""
    mk = open(os.path.join(CSRC, "Makefile")).read()
    flags = re.search(r"^FLAGS\s*:=\s*(.*)$", mk, re.M).group(1).split()
    hip = re.search(r"^HIPFLAGS\s*:=\s*(.*)$", mk, re.M).group(1)
    extra = [t for t in hip.replace("$(FLAGS)", "").replace("--offload-arch=$(ARCH)", "").split()]
    if unit_a:
        extra += re.search(r"^HIPFLAGS_A\s*:=\s*(.*)$", mk, re.M).group(1).replace("$(HIPFLAGS)", "").split()
    else:
        extra += ["-DPT_WF_TU_B"]
    return [f for f in flags if f != "-Wall"] + extra


def _resources(src, unit_a=False):
    out = subprocess.run(["hipcc", "--offload-arch=gfx950", "--cuda-device-only", "-c", src, "-o", os.devnull,
                          "-Rpass-analysis=kernel-resource-usage"] + _device_flags(unit_a), cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    res, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = res.setdefault(m.group(1), {})
            continue
        for key, name in (("VGPRs:", "vgprs"), ("ScratchSize", "scratch"), ("Occupancy", "occupancy"), ("VGPRs Spill", "vgpr_spill"), ("LDS Size", "lds")):
            m = re.search(re.escape(key) + r"[^0-9]*(\d+)", line)
            if m and cur is not None and key in line:
                cur[name] = int(m.group(1))
    return res


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not installed")
def test_trace_and_shade_kernels_keep_their_register_budget():
    res = _resources("pt_wavefront.hip", unit_a=True)             # refill trace + shade of the default schedule
    res_b = _resources("pt_wavefront.hip")                        # HAS_TLAS (and schedules 0, 2, 3, 4)

    def pick(*parts, res=res):
        hits = [v for k, v in res.items() if all(p in k for p in parts)]
        assert hits, (parts, sorted(res))
        return hits

    # refill trace kernel, no-statistics instantiations (the ones a render uses): main and tail launch, 64- and 128-slot ranges
    for r in pick("pt_wf_trace_refillILb0E"):
        assert r["scratch"] == 0 and r["vgpr_spill"] == 0, r
        assert r["occupancy"] == 8 and r["vgprs"] <= 64, r
        assert r["lds"] <= 5120, r                      # 32 one-wave workgroups per CU in 160 KB
    # shade kernel: four waves per SIMD, no scratch
    for r in pick("pt_wf_shadeILb0E"):
        assert r["scratch"] == 0 and r["vgprs"] <= 128 and r["occupancy"] >= 4, r
    # HAS_TLAS refill kernel: six waves per SIMD, no scratch
    for r in pick("pt_wf_trace_refill_tlasILb0E", res=res_b):
        assert r["scratch"] == 0 and r["vgprs"] <= 80 and r["occupancy"] >= 6, r
        assert r["lds"] * 24 <= 160 * 1024, r
