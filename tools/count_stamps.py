"""Diagnostic: shader cycles per phase of count_buckets_kernel (library built with -DBL_COUNT_STAMPS as biolib_amd/lib/libbiolib_amd_stamps.so)."""
import os, sys, ctypes as C
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
os.environ["BIOLIB_AMD_LIB"]=os.path.join(ROOT,"biolib_amd/lib/libbiolib_amd_stamps.so")
sys.path.insert(0,ROOT)
import torch, biolib_amd as B
from biolib_amd import capi
ctx=B.Context(0)
n=1_500_000_000//150*150
b=ctx.synth(42,n,150)
recs,h=b.super_kmer_records(31,15,seed=42,canonical=True)
keys=ctx.empty_u64(int(n*0.82)); cnts=torch.empty(int(n*0.82),dtype=torch.int32,device="cuda")
L=capi.lib()
out=(C.c_ulonglong*8)()
L.bl_dbg_count_stamps(out)
u,c=ctx.count_super_kmers(recs,31,15,seed=42,canonical=True,out=(keys,cnts))
L.bl_dbg_count_stamps(out)
v=list(out)
nb=v[7]
names=["between buckets (loop, waits for records/range)","clear+stage","size prefix","insert","occupied count","write out"]
print("buckets (both passes):",nb)
for i,nm in enumerate(names): print(f"  {nm:50s} {v[i]/nb:9.0f} cycles per bucket")
print(f"  {'sum':50s} {sum(v[:6])/nb:9.0f}")
