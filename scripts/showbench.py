"""print a bench.py JSON line compactly: python scripts/showbench.py gpurun_out/x/bench.json"""
import json, sys
d = json.load(open(sys.argv[1]))
c = d.pop('configs', {}); cb = d.pop('cpu_baseline', None)
print('HEAD value', d['value'], 'ms/step', d['ms_per_step'], 'enc', d['encode_GBps'], 'dec', d['decode_GBps'], 'roof', d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('traffic'))
print('    ', d['kernel_ms_per_step'])
for k, v in c.items():
    print(k, {a: b for a, b in v.items() if a not in ('workload', 'parity', 'kernel_ms_per_step', 'roofline', 'value_note', 'steps', 'blocks_per_gpu')})
    if 'kernel_ms_per_step' in v: print('    ', v['kernel_ms_per_step'], v['roofline']['kernel'], v['roofline']['frac'])
if cb: print(cb)
