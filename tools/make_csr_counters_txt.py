"""profiles/r05_csr_counters.json (tools/run_pmc_csr_r05.sh) + profiles/r02_csr_hex27_rowblock_counters.json -> profiles/r05_csr_counters.txt"""
import json, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r5 = json.load(open(f"{R}/profiles/r05_csr_counters.json"))
r2 = json.load(open(f"{R}/profiles/r02_csr_hex27_rowblock_counters.json"))
r2k = [v for k, v in r2.items() if "csr_rb" in k][0]
r2k = {c: (x["mean"] if isinstance(x, dict) else x) for c, x in r2k.items()}
out = []
P = out.append
P("SQ / TA / TCP / TCC counters of the CSR kernels behind mul! (rocprofv3 --pmc, one pass per group: tools/run_pmc_csr_r05.sh; sums over the chip, mean per launch),")
P("round 5's kernels next to round 2's first row-block kernel (profiles/r02_csr_hex27_rowblock_counters.json).  Cycles: GRBM_GUI_ACTIVE / 8 XCDs at 2.4 GHz.\n")
def row(name, d):
    cyc = d["GRBM_GUI_ACTIVE"] / 8
    f = lambda k: d.get(k, float("nan"))
    P(name)
    P(f"    kernel time under counters        {cyc / 2.4e6:8.3f} ms")
    P(f"    TA busy (TA_BUSY_avr / cycles)    {f('TA_BUSY_avr') / cyc:8.2f}")
    P(f"    VMEM read instructions            {f('SQ_INSTS_VMEM_RD') / 1e6:8.1f} M   LDS {f('SQ_INSTS_LDS') / 1e6:.1f} M   VALU {f('SQ_INSTS_VALU') / 1e6:.1f} M   SALU {f('SQ_INSTS_SALU') / 1e6:.1f} M")
    P(f"    L1 (TCP) tag accesses             {f('TCP_TOTAL_CACHE_ACCESSES_sum') / 1e6:8.1f} M = {f('TCP_TOTAL_CACHE_ACCESSES_sum') / f('SQ_INSTS_VMEM_RD'):.1f} per VMEM instruction; {f('TCP_TOTAL_CACHE_ACCESSES_sum') / 256 / cyc:.2f} per CU and cycle")
    P(f"    L1 -> L2 read requests            {f('TCP_TCC_READ_REQ_sum') / 1e6:8.1f} M   L2 requests {f('TCC_REQ_sum') / 1e6:.1f} M, hits {f('TCC_HIT_sum') / 1e6:.1f} M, misses {f('TCC_MISS_sum') / 1e6:.1f} M")
    P(f"    HBM: FETCH_SIZE x 2 x 1 KB + WRITE {(f('FETCH_SIZE') * 2 * 1024 + f('WRITE_SIZE') * 1024) / 1e9:8.3f} GB")
    if "SQ_WAIT_ANY" in d:
        P(f"    wave cycles in s_waitcnt (SQ_WAIT_ANY / SQ_WAVE_CYCLES) {f('SQ_WAIT_ANY') / f('SQ_WAVE_CYCLES'):.2f}; issuing VALU {f('SQ_ACTIVE_INST_VALU') / f('SQ_WAVE_CYCLES'):.2f}; LDS {f('SQ_ACTIVE_INST_LDS') / f('SQ_WAVE_CYCLES'):.2f}")
    else:
        P(f"    wave cycles waiting on an instruction (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES) {f('SQ_WAIT_INST_ANY') / f('SQ_WAVE_CYCLES'):.2f}; any instruction active {f('SQ_ACTIVE_INST_ANY') / f('SQ_WAVE_CYCLES'):.2f}")
    P("")
row("ROUND 2  hex-27 128^3 (C4 matrix), k_spmv_csr_rb<long,1536,16>, first version (no parity passes' lane packing, no column elision)", r2k)
names = {"c4": "ROUND 5  hex-27 128^3 (C4 matrix)", "c3": "ROUND 5  hex-8 elasticity 128^3 (C3 matrix, 81-entry rows)", "c2": "ROUND 5  hex-8 thermal 256^3 (C2 matrix)"}
for leg in ("c4", "c3", "c2"):
    for k, v in r5[leg].items():
        row(names[leg] + ", " + k[:44], v)
P("""Reading.  Row-block kernel, C4: against round 2 the kernel issues 18 % fewer memory instructions and 16 % fewer L1 tag accesses for the same matrix (column elision: the
column stream of a tile whose rows repeat their first two rows' offsets is not read) and the texture addresser went from 66 % to 57 % busy; 62 % of the wave cycles still sit in
s_waitcnt with the addresser a little over half busy and HBM a little over half busy (10.5 GB in 2.5 ms): neither unit is saturated, the waves alternate between the two
(8 one-wave workgroups per CU, one tile in flight each).  What the gathers cost: ~22 L1 tag accesses per memory instruction on C3 / C4 against ~19.6 on the hex-8 matrix whose
gathers are unit-stride across lanes (rows = lanes); 0.44-0.48 tag accesses per CU and cycle on the two row-block matrices -- under half of what the L1 can look up.
(The hex-8 leg has five launches only, the first ones cold: its time under counters is above that of a warm launch; the ratios per instruction stand.)
The x-window variant (profiles/r05_csr_rb_xwin.txt) removed the gathers' addresser work and lost more to the exposed round trip per tile than it gained.
Late in round 5 the ISA of k_spmv_csr_w showed why its software pipeline did not overlap what it was built to overlap: wherever paths with different numbers of loads meet
(a request behind `if (t_next < ntiles)`, a column stream behind `if (!el)`, a gather behind `if (j < hi)`), the compiler's s_waitcnt for "the gathers have returned" must assume the
smallest count -- vmcnt(0): the row sums waited for the next tile's streams too.  With every load issued on every path (scalar loads for tile-level values, a bounded column
buffer for elided tiles, inactive lanes gathering x[0]) the waits count down (vmcnt(57) ... vmcnt(30)) as intended: mul! 8.1-8.5 -> 7.9-8.1 ms at 512^3, 1.03-1.04 -> 1.02 ms at 256^3
-- the kernel is bound by the loaded memory latency at eight waves per CU (~10 us per tile and wave), not by the order of its waits.  The same change in k_spmv_csr_rb was
measured and not kept: its chunks of 16 gathers are mostly empty on short rows, issuing all of them cost 12 %; with the gathers left conditional nothing changed.
Two tiles in flight per wave in k_spmv_csr_w (a second register set for the tile after next: 43 KB per wave under way instead of 21.5; 256 VGPRs + 120 bytes of scratch) was
measured too: 1.29 ms at 256^3 and 9.8 ms at 512^3 against 1.02 / 7.9 -- more requests in flight only lengthen the queues: at 5.0 TB/s of counter traffic the kernel sits at
80 % of what a streaming kernel reaches on this chip (6.3 TB/s), and the rest of its requests are the gathers' L2 traffic.  Not kept.""")
open(f"{R}/profiles/r05_csr_counters.txt", "w").write("\n".join(out) + "\n")
