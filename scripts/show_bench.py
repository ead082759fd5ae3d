import json, sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; d=json.loads(l)
r=d['roofline']
def rf(r):
    h=r.get('hbm'); v=r.get('valu')
    return "bound %s frac %s | hbm %s B/ray %s L2hit %s | valu busy %s lane %s | alg B/ray %s" % (r.get('bound'), r.get('frac'), h and h['frac'], h and h['bytes_per_ray'], h and h['l2_hit_rate'], v and v['busy_frac'], v and v['lane_use'], r['algorithmic']['bytes_per_ray'])
print("C2 samples/s %.4g  image mean %.6f (a variant that changes it is WRONG)  ms/step %.1f  k_extend ms/launch %.3f | %s" % (d['value'], d['image_mean'], d['ms_per_step'], r['avg_launch_ms'], rf(r)))
for k in ('coherent','incoherent'):
    if 'closest_hit' in d:
        c=d['closest_hit'][k]; print("%-10s %7.1f Mrays/s | %s | %s" % (k, c['Mrays_per_s'], rf(c['roofline']), c.get('cpu_oracle')))
