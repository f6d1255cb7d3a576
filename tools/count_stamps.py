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
print("buckets(both passes)",v[5],"cycles/bucket: clear",v[0]/v[5],"load+scan",v[1]/v[5],"insert",v[2]/v[5],"occ-scan",v[3]/v[5],"whole",v[4]/v[5])
