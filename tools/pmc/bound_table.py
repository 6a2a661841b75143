"""One line per (segment, kernel) of tools/pmc/bound_summary.py's output with the derived quantities that say what bounds a kernel:
python3 tools/pmc/bound_table.py gpurun_out/pmc_bound/summary.txt   (development tool)"""
import re, sys
seg = None; rows = []
for line in open(sys.argv[1]):
    m = re.match(r"segment\s+(\d+) level (\d+) op (\d+) kernel-kind (\d+) bytes (\d+)\s+(\S.*?)\s+event-timed ([\d.]+) us", line)
    if m:
        seg = {"seg": int(m.group(1)), "level": int(m.group(2)), "op": int(m.group(3)), "bytes": int(m.group(5)), "kernel": m.group(6), "us": float(m.group(7)), "c": {}}
        rows.append(seg); continue
    m = re.match(r"\s+(\S+)\s+n=\s*\d+ mean=\s*([\d.]+)", line)
    if m and seg: seg["c"][m.group(1)] = float(m.group(2))
print("level op kernel                     us(cold) prof_us | wait%  vmem/wave-inst lvl | tcp_lines  L1miss%  L1->L2 lat  pend_stall%  ta_data_stall% | L2 req   hit%  tag_stall% busy% | EA rd  lat(cyc)  dram_credit_stall%")
for s in rows:
    c = s["c"]; g = lambda k: c.get(k, float("nan"))
    wave = g("SQ_WAVE_CYCLES"); gate = g("TCP_GATE_EN1_sum"); cyc = g("TCC_CYCLE_sum")
    print(f"{s['level']:3d} {s['op']:2d} {s['kernel'][:26]:26s} {s['us']:8.1f} {g('_us_profiled'):7.1f} | "
          f"{100*g('SQ_WAIT_ANY')/wave:5.1f} {g('SQ_INSTS_VMEM_RD')/1e6:7.2f}M {g('SQ_INST_LEVEL_VMEM')/max(g('SQ_BUSY_CYCLES'),1):6.1f} | "
          f"{g('TCP_TOTAL_CACHE_ACCESSES_sum')/1e6:8.2f}M {100*g('TCP_TCC_READ_REQ_sum')/max(g('TCP_TOTAL_CACHE_ACCESSES_sum'),1):6.1f} {g('TCP_TCC_READ_REQ_LATENCY_sum')/max(g('TCP_TCC_READ_REQ_sum'),1):9.0f} "
          f"{100*g('TCP_PENDING_STALL_CYCLES_sum')/gate:10.1f} {100*g('TCP_TCP_TA_DATA_STALL_CYCLES_sum')/gate:12.1f} | "
          f"{g('TCC_REQ_sum')/1e6:7.2f}M {100*g('TCC_HIT_sum')/max(g('TCC_HIT_sum')+g('TCC_MISS_sum'),1):5.1f} {100*g('TCC_TAG_STALL_sum')/cyc:8.2f} {100*g('TCC_BUSY_sum')/cyc:6.1f} | "
          f"{g('TCC_EA0_RDREQ_sum')/1e6:6.2f}M {g('TCC_EA0_RDREQ_LEVEL_sum')/max(g('TCC_EA0_RDREQ_sum'),1):7.0f} {100*g('TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum')/cyc:8.1f}")
