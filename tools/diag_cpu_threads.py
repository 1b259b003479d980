"""Diagnostic (GPU box host): oracle throughput vs thread count (sizing of bench.py's cpu_baseline leg)."""
import sys, time
sys.path.insert(0,'/root/repo')
import __graft_entry__ as g
O=g.load_oracle()
sc=O.cornell_box(1024,1024)
for th in (8,16,32,64,256):
    t=time.perf_counter(); m,v,st=O.render(sc,256,max_depth=8,region=(0,504,1024,520),threads=th,want_stats=True); dt=time.perf_counter()-t
    print(th, st['samples']/dt/1e6, flush=True)
