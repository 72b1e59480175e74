#!/bin/bash
# round 5, session 10: (a) 8192 points: the frames dispatched ahead of the candidate lists (ordfe) against the order that ships (base),
# alternating runs of 200 and of 20 steps; (b) the driver's form of the default line with this round's also entries
OUT=gpurun_out/r05_s10
mkdir -p $OUT
R=/root/repo
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 PYTHONFAULTHANDLER=1 TMPDIR=/tmp
SECONDS=0; timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default_k20.json 2> $OUT/bench_default_k20.err
echo "default bench: $SECONDS s"; tail -5 $OUT/bench_default_k20.err | cut -c1-300
python - <<'PY'
import json
j = json.loads(open('gpurun_out/r05_s10/bench_default_k20.json').read().strip().splitlines()[-1])
print('default k20', j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline'].get('traffic_over_algorithmic'))
for a in j.get('also', []):
    if a.get('variant') == 'drop_in_path':
        print('   drop-in', a.get('fft_size'), a.get('ss_process_MSps'), a.get('ss_feed_MSps'), a.get('pcie_bound_MSps'), a.get('error'))
        continue
    print('   also', a.get('variant'), a.get('baseline_config'), a.get('frames_per_batch'), a.get('ms_per_step'), a.get('value'), a.get('error'),
          [(k['slot'], k['us'], k['frames_per_launch'], k['frac_of_peak']) for k in a.get('kernels', [])])
    p = a.get('parity') or {}
    print('        parity', p.get('failed'), p.get('timed_path'), (p.get('all_bins_vs_fp64_fft_dB') or {}).get('engine_over_reference_rms'))
p = j.get('parity') or {}
print('parity', p.get('failed'), p.get('all_bins_vs_fp64_fft_dB'))
print('cpu', (j.get('cpu_baseline') or {}).get('value'))
PY
