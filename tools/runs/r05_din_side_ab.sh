# DIN train step at B 4096: one stream against the two-stream merge schedule (REC_DIN_SIDE), mirror and C entry
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
python - <<'PY'
import os, sys, torch
sys.path.insert(0, '.')
from paddlerec_amd.din import DINLayer
DEV="cuda"
def run(B,T,side,entry):
    os.environ["REC_DIN_SIDE"]=side
    torch.manual_seed(5)
    g = torch.Generator(device=DEV).manual_seed(3)
    m = DINLayer(64, 64, "sigmoid", False, True, 63001, 801, device=DEV)
    hi = torch.randint(0, 63001, (B, T), device=DEV, generator=g)
    hc = torch.randint(0, 801, (B, T), device=DEV, generator=g)
    ti = torch.randint(0, 63001, (B, 1), device=DEV, generator=g)
    tc = torch.randint(0, 801, (B, 1), device=DEV, generator=g)
    lens = torch.randint(1, T + 1, (B, 1), device=DEV, generator=g)
    mask = torch.where(torch.arange(T, device=DEV)[None] < lens, 0, -1000000000).long()
    label = (torch.rand(B, 1, device=DEV, generator=g) < 0.5).float()
    tis, tcs = ti.expand(B, T).contiguous(), tc.expand(B, T).contiguous()
    a=(hi,hc,ti,tc,label,mask,tis,tcs)
    step=getattr(m, entry)
    for _ in range(10): step(*a)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): l,_p=step(*a)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/50, float(l), m
for B,T in ((4096,100),(4096,512)):
    t0,l0,m0=run(B,T,"0","train_step")
    for side,entry in (("1","train_step"),("0","train_step_c"),("1","train_step_c")):
        t1,l1,m1=run(B,T,side,entry)
        same=all(torch.equal(v, m1.state_dict()[k]) for k,v in m0.state_dict().items())
        print("B %d T %d  mirror one stream %.3f ms | %s side=%s %.3f ms  loss %.6f %.6f  params identical %s"%(B,T,t0,entry,side,t1,l0,l1,same))
PY
