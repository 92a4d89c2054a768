import sys, json
for f in sys.argv[1:]:
    for l in open(f):
        if l.startswith('{'):
            j = json.loads(l); r = j['roofline']
            print('%-40s value %.3e ms/step %.4f launch %.1f us achieved %.0f GB/s frac %.3f spl %s' % (f.split('/')[-1], j['value'], j['ms_per_step'], r['avg_launch_us'], r['achieved'], r['frac'], r.get('sweeps_per_launch')))
