#!/usr/bin/env python3
"""Print the beta_coreset block of a bench.py JSON line compactly:  python tools/show_beta.py gpurun_out/x.json"""
import json
import sys
j = json.load(open(sys.argv[1]))
print('it/s %.0f  ms/step %.4f  K3 frac %.3f  init %.1f ms  K1 %.3f ms (%.3f hbm)' % (
    j['value'], j['ms_per_step'], j['roofline']['frac'], j['solver_init_ms'], j['projection']['kernel_ms'],
    j['projection']['roofline_hbm']['frac']))
print('init', json.dumps(j.get('solver_init')))
print('K4 main', json.dumps(j.get('posterior_gram')))
for e in j.get('beta_coreset', []):
    if 'cpu_baseline' in e:
        print('cpu', e['cpu_baseline'])
        continue
    b = e['breakdown_ms']
    ks = list(b.keys())
    print('%-38s M=%3d  %.3f ms/grad  K1 %.3f (hbm %.3f mfma %.3f)  non-K1 %.2f | sampler %.3f call %.3f adam %.3f | %s | mat %.3f  build %.1f' % (
        e['config'], e['M'], e['ms_per_gradient'], e['k1_store_free_kernel_ms'], e['roofline_hbm']['frac'],
        e['roofline_fp64_mfma']['frac'], e['non_k1_fraction'], b[ks[0]], b[ks[1]], b[ks[2]],
        ' '.join('%s %.3f' % (k[:9], v) for k, v in b[ks[3]].items()), e['materialising_path_ms_per_gradient'],
        [v for k, v in e.items() if k.startswith('build_step')][0]))
for e in j.get('other_configs', []):
    print('%-50s %-40s %.3f ms  hbm %.3f  mfma %.3f' % (e['config'], e['model'][:40], e['kernel_ms'], e['roofline_hbm']['frac'], e['roofline_fp64_mfma']['frac']))
    if 'posterior_gram_K4' in e:
        print('   K4', json.dumps(e['posterior_gram_K4']))
