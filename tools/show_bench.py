"""print the headline numbers of a bench.py log (the one JSON line): python tools/show_bench.py gpurun_out/b1.log"""
import json
import sys
for path in sys.argv[1:]:
    for line in open(path):
        if line.startswith('{'):
            d = json.loads(line)
            e = d.get('experiment_1_protocol', {})
            print(path, d['value'], d['ms_per_step'], 'protocol', e.get('QPs_per_sec_median'), e.get('median_forward_ms'), e.get('median_backward_ms'),
                  'first_use', d['config'].get('first_use_ms_per_step'))
            print('  kernels', d.get('kernel_ms_per_step'))
            print('  sync', d.get('step_sync_default', {}).get('ms_per_step'), 'lu', d.get('step_linsolve_lu', {}).get('ms_per_step'),
                  {k: v.get('ms') for k, v in d.get('other_configs_forward_only', {}).items()},
                  {k: v.get('ms_per_step') for k, v in d.get('other_workloads_fwd_bwd', {}).items()})
