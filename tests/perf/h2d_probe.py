"""Host <-> device copy rates of this box (page-locked and pageable): what the ingest paths can hope for per byte they move."""
import torch, time
x = torch.empty(256<<20, dtype=torch.uint8).pin_memory()
y = torch.empty(256<<20, dtype=torch.uint8, device="cuda")
for _ in range(2): y.copy_(x, non_blocking=True); torch.cuda.synchronize()
t0=time.perf_counter()
for _ in range(5): y.copy_(x, non_blocking=True)
torch.cuda.synchronize(); dt=time.perf_counter()-t0
print("pinned H2D GB/s", 5*x.numel()/dt/1e9)
z = torch.empty(256<<20, dtype=torch.uint8)
t0=time.perf_counter()
for _ in range(3): y.copy_(z)
torch.cuda.synchronize(); dt=time.perf_counter()-t0
print("pageable H2D GB/s", 3*z.numel()/dt/1e9)
t0=time.perf_counter()
for _ in range(5): x.copy_(y, non_blocking=True)
torch.cuda.synchronize(); dt=time.perf_counter()-t0
print("pinned D2H GB/s", 5*x.numel()/dt/1e9)
