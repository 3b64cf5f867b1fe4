"""The chained scan of quiet ticks (k_scan<..., CHAIN>) waits, resident, for workgroups of the launch before it: that is free of
deadlock only while TWO of its 1 024-thread workgroups fit a CU — 8 waves per SIMD, i.e. at most 64 vector registers and at most
80 scalar registers per wave (800 per SIMD, allocated in sixteens plus sixteen: MI355X_MICROARCH.md "Residency").  The compiler's
own resource report of every chained instantiation is checked here, without a GPU; the library asks the occupancy query again
when a handle is created and every in-kernel wait is bounded."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_chained_scan_fits_twice_on_a_cu(tmp_path):
    src = os.path.join(ROOT, "rio-rs_amd", "csrc", "placement_kernels.hip")
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-c", src,
                        "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", str(tmp_path / "pk.o")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    recs, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = recs.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z /\[\]]+?): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    # k_scan<VIRT, ALLALIVE, TPI, COMPACT, NT, CHAIN = true>
    chained = {k: v for k, v in recs.items() if re.match(r"_ZN5riogp6k_scanILb0ELb[01]ELi[12]ELi0ELb[01]ELb1EEE", k)}
    assert len(chained) == 8, sorted(recs)[:5]          # ALLALIVE x TPI 1 | 2 x NT
    for name, u in chained.items():
        assert u["VGPRs"] + u.get("AGPRs", 0) <= 64, (name, u)
        assert u["TotalSGPRs"] <= 80, (name, u)
        assert u["Occupancy [waves/SIMD]"] == 8, (name, u)
        if "ELi1ELi0" in name:                           # the form the product launches (one tile per wave-iteration)
            assert u["ScratchSize [bytes/lane]"] == 0, (name, u)
