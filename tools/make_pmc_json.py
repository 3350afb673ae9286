#!/usr/bin/env python
"""profiles/r0N_pmc.json from the PMC summaries of tools/gpu_profile_r0N.sh: HBM traffic and executed instructions of g2p2g_kernel<2> per
launch of C3, for the default (rest) window and the flow window.  FETCH_SIZE under-reports coalesced reads on gfx950
(MI355X_MICROARCH.md, HBM section): it is calibrated on carry_grid_kernel of the SAME pass, which reads a known byte count (every old
neighbour block once, 1 KiB each: WRITE_SIZE of the same kernel - exact, factor 1.0 - gives the block count); WRITE_SIZE is used as reported.
usage: make_pmc_json.py gpurun_out/prof_r04 profiles/r04_pmc.json"""
import json
import re
import sys


def counters(path):
    out = {}
    for line in open(path):
        m = re.match(r"\s*(\S.*?)\s+(\w+)\s+n=\s*(\d+) sum=(\S+) avg=(\S+)", line)
        if m:
            out.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(5))
    return out


def window(path, particles):
    c = counters(path)
    g = next(v for k, v in c.items() if "g2p2g" in k)
    carry = next(v for k, v in c.items() if "carry_grid" in k)
    calib = carry["WRITE_SIZE"] / carry["FETCH_SIZE"]           # KiB written (= blocks carried) / KiB the fetch counter saw for the same blocks
    read = g["FETCH_SIZE"] * 1024 * calib
    write = g["WRITE_SIZE"] * 1024
    raw = g["FETCH_SIZE"] * 1024
    return {"fetch_size_kib": g["FETCH_SIZE"], "write_size_kib": g["WRITE_SIZE"], "fetch_calibration": calib, "read_bytes": int(read), "read_bytes_uncalibrated": int(raw), "write_bytes": int(write),
            "traffic_bytes": int(read + write), "traffic_bytes_low": int(raw + write), "algorithmic_bytes": particles * 144, "valu_insts": g.get("SQ_INSTS_VALU"), "salu_insts": g.get("SQ_INSTS_SALU"),
            "lds_insts": g.get("SQ_INSTS_LDS"), "wave_cycles": g.get("SQ_WAVE_CYCLES"), "wait_any": g.get("SQ_WAIT_ANY"), "wait_inst_any": g.get("SQ_WAIT_INST_ANY"),
            "active_inst_any": g.get("SQ_ACTIVE_INST_ANY"), "lds_bank_conflict": g.get("SQ_LDS_BANK_CONFLICT"), "lds_idx_active": g.get("SQ_LDS_IDX_ACTIVE"), "active_inst_lds": g.get("SQ_ACTIVE_INST_LDS"), "wait_inst_lds": g.get("SQ_WAIT_INST_LDS")}


def main(src, dst):
    n = 40108032
    first = open(f"{src}/c3_default_pmc.txt").readline()
    stamp = first[first.rfind("(") + 1:first.rfind(")")] if "(" in first else "unstamped"
    out = {"_comment": "the sand G2P2G kernel (see `kernel`) per launch on C3 (40 108 032 sand particles, 512^3), MI355X; rocprofv3 --pmc passes of tools/gpu_profile_rNN.sh "
                       "(profiles/rNN_c3_default_pmc.txt, rNN_c3_moving_pmc.txt of the same round as this file); FETCH_SIZE calibrated on carry_grid_kernel of the same pass (a streaming kernel: the counter tallies a "
                       "128-B request as 64 B), WRITE_SIZE as reported.  The calibration holds for the rest window, whose record reads are streams; the flow window reads scattered 32-B records and row entries "
                       "(64-B requests are tallied in full), so its true read volume lies between read_bytes_uncalibrated and read_bytes: traffic_bytes is an UPPER bound there, traffic_bytes_low the lower one",
           "kernel": "g2p2g_pair_kernel<2> (g2p2g_kernel<2> before round 6)", "particles": n, "stamp": stamp,
           "rest": window(f"{src}/c3_default_pmc.txt", n), "flow": window(f"{src}/c3_moving_pmc.txt", n)}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
