import json, sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]; d=json.loads(l)
r=d['roofline']
print("C2 samples/s %.3g  Mrays/s(kernel) %.1f  frac %.3f  ms/step %.1f" % (d['value'], r['Mrays_per_s_in_kernel'], r['frac'], d['ms_per_step']))
for k in ('coherent','incoherent'):
    if 'closest_hit' in d:
        c=d['closest_hit'][k]; print(k, c['Mrays_per_s'], c['roofline']['frac'], 'culled', c.get('culled_candidate_frac'), c.get('cpu_oracle'))
