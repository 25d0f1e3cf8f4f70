#!/bin/bash
# Runs on the GPU box (gpurun): memory-system counters per kernel of bench.py (L1 / L2 / fabric latencies, hit rates,
# address-translation hits and misses, stalls) in four --pmc passes (kernel trace only)
#   tools/profile_mem.sh <tag>
TAG=${1:-r03_mem}; shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ring --no-extras"
pass() {  # name counters...   (a counter set the hardware cannot collect in one pass makes rocprofv3 abort: bounded, reported)
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/$name -o run --output-format csv -- $B > /dev/null 2> $OUT/$name.err \
    && echo "mem pass $name done" || { echo "mem pass $name FAILED: $(grep -m1 -i 'exceeds\|error' $OUT/$name.err | cut -c1-160)"; }
}
pass sq1 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
pass sq2 TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
pass sq3 TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_CYCLE_sum
pass sq4 TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
python3 - $OUT $TAG <<'PY'
import sys, os, json
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'tools'))
import pmc_summary as ps
out, tag = sys.argv[1], sys.argv[2]
res = {}
for d in ('sq1', 'sq2', 'sq3', 'sq4'):
    for k, counters in ps.all_counters(os.path.join(out, d)).items():
        if not (k.startswith('bev_') or k.startswith('k1_')):
            continue
        for name, v in counters.items():
            tail = v[len(v) // 2:]
            res.setdefault(k, {})[name] = round(sum(tail) / len(tail))
json.dump({'note': 'rocprofv3 --pmc memory-system counters (three passes, --kernel-trace only), per launch, bench.py steady state', 'kernels': res},
          open(os.path.join(out, f'{tag}_pmc_mem_per_kernel_avg.json'), 'w'), indent=1)
for k, c in res.items():
    g = lambda n: c.get(n, 0)
    print(k)
    print('   L1->L2 read latency %.0f cycles (%d reqs) | L2 hit rate %.2f (%d hits, %d misses) | EA read latency %.0f cycles (%d reqs) | EA write latency %.0f (%d)' % (
        g('TCP_TCC_READ_REQ_LATENCY_sum') / max(g('TCP_TCC_READ_REQ_sum'), 1), g('TCP_TCC_READ_REQ_sum'), g('TCC_HIT_sum') / max(g('TCC_HIT_sum') + g('TCC_MISS_sum'), 1),
        g('TCC_HIT_sum'), g('TCC_MISS_sum'), g('TCC_EA0_RDREQ_LEVEL_sum') / max(g('TCC_EA0_RDREQ_sum'), 1), g('TCC_EA0_RDREQ_sum'),
        g('TCC_EA0_WRREQ_LEVEL_sum') / max(g('TCC_EA0_WRREQ_sum'), 1), g('TCC_EA0_WRREQ_sum')))
    print('   UTCL1 requests %d hits %d misses %d | TCP pending stall %d, TA data stall %d, TCR stall %d of gate cycles %d | L2 busy avr %s, TA busy avr %s | EA rd credit stall %d, wr stall %d, too many wr %d, tag stall %d of L2 cycles %d' % (
        g('TCP_UTCL1_REQUEST_sum'), g('TCP_UTCL1_TRANSLATION_HIT_sum'), g('TCP_UTCL1_TRANSLATION_MISS_sum'), g('TCP_PENDING_STALL_CYCLES_sum'), g('TCP_TCP_TA_DATA_STALL_CYCLES_sum'),
        g('TCP_TCR_TCP_STALL_CYCLES_sum'), g('TCP_GATE_EN1_sum'), g('TCC_BUSY_avr'), g('TA_BUSY_avr'), g('TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum'), g('TCC_EA0_WRREQ_STALL_sum'),
        g('TCC_TOO_MANY_EA_WRREQS_STALL_sum'), g('TCC_TAG_STALL_sum'), g('TCC_CYCLE_sum')))
PY
