import os, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
os.environ["FLBGPU_DEBUG"]="1"
import flbamd_loader, synth
import oracle_binding as ob
from bench import APACHE2, TIME_FMT, GREP_RULE
g=flbamd_loader.load(); g.init(0)
n=int(sys.argv[1]) if len(sys.argv)>1 else 1000
data,off,ep=synth.apache_records(n)
blob=bytes(data)
p=g.Parser(APACHE2,time_fmt=TIME_FMT,time_key="time")
fp=g.FilterParser("log",[p]); fg=g.FilterGrep([GREP_RULE])
ch=g.FilterChain([fp,fg])
print("running", flush=True)
r,o=ch.filter(blob)
print("ret",r,len(o) if o else o, ch.last_stats(), flush=True)
po=ob.Parser(APACHE2,time_fmt=TIME_FMT,time_key="time")
r1,o1=ob.FilterParser("log",[po]).filter(blob); r2,o2=ob.Grep([GREP_RULE]).filter(o1)
print("oracle",r2,len(o2), o==o2)
