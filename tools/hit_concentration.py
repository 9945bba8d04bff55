"""Which filters the hits of BASELINE config 3 come from (CPU only: generator + oracle on the full 10 M-subscription table,
first 3 000 publish topics).  Output: profiles/r02n_config3_hit_concentration_oracle.txt.  python tools/hit_concentration.py"""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import oracle as orc
W = bench.gen_workload(3, 1.0)
blob, offs = W['blob'], W['offs']
t=time.time()
mv = memoryview(np.ascontiguousarray(blob)).cast('B')
d = {}
fid = np.empty(len(offs)-1, dtype=np.int32)
o = offs.tolist()
bb = bytes(mv)
for i in range(len(o)-1):
    k = bb[o[i]:o[i+1]]
    v = d.get(k)
    if v is None:
        v = len(d); d[k] = v
    fid[i] = v
print('distinct filters', len(d), round(time.time()-t,1), flush=True)
names = [None]*len(d)
for k,v in d.items(): names[v]=k
t=time.time()
r = orc.DefaultRouter(); r.add_bulk(blob, offs, W['client'], W['qos'])
print('oracle built', round(time.time()-t,1), flush=True)
n=3000
sb, so = bench.prefix(W, n)
res = r.match_flat(sb, so)
sub_ids = res['sub_ids']
print('hits', len(sub_ids))
cnt = np.bincount(fid[sub_ids], minlength=len(d)).astype(np.int64)
order = np.argsort(-cnt)
tot = cnt.sum(); cum = np.cumsum(cnt[order])
for k in (1,2,4,8,16,32,64,128,256,1024,4096):
    print(f'top {k} filters: {cum[k-1]/tot*100:.1f} % of hits')
print('filters with any hit in the sample:', int((cnt>0).sum()))
for j in order[:12]: print(names[j].decode(), cnt[j], 'subs of this filter:', int((fid==j).sum()))
# v5 relevance: how many DISTINCT clients per hot filter pair overlap? (clients subscribed to both of the two hottest filters)
cl = W['client']
a, b = order[0], order[1]
ca, cb = set(cl[fid==a].tolist()), set(cl[fid==b].tolist())
print('clients in hottest', len(ca), 'second', len(cb), 'both', len(ca & cb))
