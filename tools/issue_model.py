#!/usr/bin/env python3
"""What the instruction stream of every kernel costs to ISSUE, in absolute time, next to what the kernel took.

    python tools/issue_model.py profiles/r05_final            (reads pmc_sq_l2.csv and kernel_stats.csv of a tools/collect_profiles.sh collection)

predicted_issue_ms = sum over instruction classes (dynamic count per wavefront x the class's MEASURED issue cost) x wavefronts per SIMD / clock

  * dynamic counts: rocprofv3 --pmc SQ_INSTS_VALU (all VALU) and its classes SQ_INSTS_VALU_{ADD,MUL,FMA}_F32, _TRANS_F32, _INT32, _CVT, per launch,
    divided by SQ_WAVES (one pixel per lane: per wavefront = per 64 pixels);
  * issue cost per wave64 instruction per SIMD (tools/microbench/valu_rates2.hip on this part, profiles/r03_microbench/valu_rates2.txt, 8 waves per
    SIMD): fp32 add / mul / fma 2.7 cycles, transcendentals 8.3, conversions 4.2; integer ops and everything the class counters do not name
    ("other": compares and moves 2.7, min / max / med3, cndmask behind its compare, fma_mix, fract 4.2, packed fp32 4.55) at the mean cost of
    THAT KERNEL's own mix of them, read from its compiled ISA (tools/isa_mix.py -> <profile>/isa_other_mix.json; without that file: 3.4 and
    4.2, the flat figures of round 4, which priced K1 — whose "other" is mostly compares and moves — above its own run time);
  * wavefronts per SIMD = SQ_WAVES / 1024; clock = (GRBM_GUI_ACTIVE / 8 XCDs) / the kernel's measured duration.

Next to it, the LDS side of the tiled kernels: lds_busy_ms = SQ_LDS_IDX_ACTIVE / 256 CUs / clock (the time the CU's one LDS pipe, which its four SIMDs
share, is occupied), of which lds_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE is spent re-issuing conflicting lanes.

A kernel whose measured time equals the prediction is bound by VALU issue alone; the gap is what its waves wait for with nothing else to issue
(memory latency the other waves of the SIMD do not cover, barriers, the tail of the launch).  This replaces rounds 2-3's `valu_busy`
(SQ_ACTIVE_INST_VALU x 4 / cycles), which counts one quad-cycle per instruction whatever the instruction costs and exceeded 1.
"""
import csv
import os
import sys

RATE = {"fma_add_mul": 2.7, "trans": 8.3, "cvt": 4.2, "int": 3.4, "other": 4.2}
N_SIMD, N_XCD, N_CU = 1024, 8, 256
# rocprofv3 kernel-name fragments -> tools/isa_mix.py's kernel keys
MIX_KEY = (("k1_ssgi_march", "k1_ssgi_march"), ("k2_temporal_reproject", "k2_temporal_reproject"), ("k3_tiled<true", "k3_poisson_denoise_pass0"),
           ("k3_tiled<false", "k3_poisson_denoise_pass1"), ("k4_compose", "k4_compose"))


def _isa_mix(profile_dir):
    import json
    try:
        return json.load(open(os.path.join(profile_dir, "isa_other_mix.json")))
    except Exception:  # noqa: BLE001
        return {}


def _counters(profile_dir):
    acc = {}
    path = os.path.join(profile_dir, "pmc_sq_l2.csv")
    if not os.path.exists(path):
        return acc
    for r in csv.DictReader(open(path)):
        acc.setdefault(r["kernel"], {})[r["counter"]] = float(r["mean_value"])
    return acc


def _durations_us(profile_dir):
    out = {}
    path = os.path.join(profile_dir, "kernel_stats.csv")
    if not os.path.exists(path):
        return out
    for r in csv.DictReader(open(path)):
        name = r.get("Name") or r.get("KernelName") or ""
        avg = r.get("AverageNs") or r.get("Average") or r.get("AvgDurationNs")
        if name and avg:
            out[name] = float(avg) / 1e3
    return out


def model(c, measured_us=None, pixels=3840 * 2160, persistent=False, rates=None):
    """c: counter name -> mean per launch.  Returns a dict, or None when the class counters are missing.
    persistent: the launch's wavefronts loop over tiles (K1 since round 4): instructions per pixel are counts / pixels, not counts / (64 x SQ_WAVES)."""
    need = ("SQ_INSTS_VALU", "SQ_WAVES", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32")
    if not all(k in c for k in need):
        return None
    w = c["SQ_WAVES"]
    fam = c["SQ_INSTS_VALU_ADD_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + c["SQ_INSTS_VALU_FMA_F32"]
    tr = c["SQ_INSTS_VALU_TRANS_F32"]
    cvt = c.get("SQ_INSTS_VALU_CVT", 0.0)
    it = c.get("SQ_INSTS_VALU_INT32", 0.0)
    other = max(c["SQ_INSTS_VALU"] - fam - tr - cvt - it, 0.0)
    r_int, r_other = (rates or {}).get("int_rate", RATE["int"]), (rates or {}).get("other_rate", RATE["other"])
    cycles_per_wave = (fam * RATE["fma_add_mul"] + tr * RATE["trans"] + cvt * RATE["cvt"] + it * r_int + other * r_other) / w
    pw = pixels / 64.0 if persistent else w  # wave64 instructions per pixel = per (wavefront of 64 pixels)
    out = {"valu_per_px": round(c["SQ_INSTS_VALU"] / pw, 1),
           "per_px": {"fp32_add_mul_fma": round(fam / pw, 1), "transcendental": round(tr / pw, 1), "cvt": round(cvt / pw, 1), "int32": round(it / pw, 1), "other": round(other / pw, 1)},
           "issue_cycles_per_wave": round(cycles_per_wave, 0), "waves_per_simd": round(w / N_SIMD, 1), "rate_int": r_int, "rate_other": r_other}
    if measured_us and "GRBM_GUI_ACTIVE" in c:
        clock_ghz = c["GRBM_GUI_ACTIVE"] / N_XCD / (measured_us * 1e3)
        pred_ms = cycles_per_wave * (w / N_SIMD) / (clock_ghz * 1e9) * 1e3
        out.update(clock_GHz=round(clock_ghz, 3), predicted_issue_ms=round(pred_ms, 4), measured_ms=round(measured_us / 1e3, 4),
                   issue_share_of_measured=round(pred_ms / (measured_us / 1e3), 3))
        if c.get("SQ_LDS_IDX_ACTIVE"):
            lds_ms = c["SQ_LDS_IDX_ACTIVE"] / N_CU / (clock_ghz * 1e9) * 1e3
            out.update(lds_busy_ms=round(lds_ms, 4), lds_busy_share_of_measured=round(lds_ms / (measured_us / 1e3), 3),
                       lds_conflict_share=round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 3))
    return out


def table(profile_dir):
    cs, du, mix = _counters(profile_dir), _durations_us(profile_dir), _isa_mix(profile_dir)
    rows = {}
    for k, c in cs.items():
        key = next((mk for frag, mk in MIX_KEY if frag in k), None)
        m = model(c, du.get(k), persistent="k1_ssgi_march" in k, rates=mix.get(key))
        if m:
            rows[k] = m
    return rows


def main():
    d = sys.argv[1] if len(sys.argv) > 1 else "profiles/r05_final"
    rows = table(d)
    if not rows:
        sys.exit("no class counters in %s/pmc_sq_l2.csv (collect with tools/collect_profiles.sh)" % d)
    print("%-52s %8s %8s %7s %6s %6s %7s %11s | %9s %9s %6s | %8s %6s %9s" % ("kernel", "VALU/px", "fma/add", "trans", "cvt", "int", "other", "int/other c",
                                                                                "issue ms", "meas. ms", "share", "LDS ms", "share", "conflicts"))
    for k, m in sorted(rows.items(), key=lambda kv: -kv[1].get("measured_ms", 0)):
        p = m["per_px"]
        print("%-52s %8.1f %8.1f %7.1f %6.1f %6.1f %7.1f %5.2f/%5.2f | %9s %9s %6s | %8s %6s %9s" % (
            k[:52], m["valu_per_px"], p["fp32_add_mul_fma"], p["transcendental"], p["cvt"], p["int32"], p["other"], m["rate_int"], m["rate_other"],
            m.get("predicted_issue_ms", "-"), m.get("measured_ms", "-"), m.get("issue_share_of_measured", "-"),
            m.get("lds_busy_ms", "-"), m.get("lds_busy_share_of_measured", "-"), m.get("lds_conflict_share", "-")))


if __name__ == "__main__":
    main()
