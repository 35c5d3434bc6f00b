import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import a2f_amd
from a2f_amd import ops
from a2f_amd.ops import MMOp
M, N, K = 600, 768, 768
g = torch.Generator().manual_seed(1)
A = torch.randn(M, K, generator=g).cuda(); B = torch.randn(N, K, generator=g).cuda()
bias = torch.randn(N, generator=g).cuda(); aux = torch.randn(M, N, generator=g).cuda()
be = ops.backend()
ref = A.double() @ B.double().t()
for name, kw in (("plain", {}), ("bias", dict(bias=bias)), ("splitk", dict(split_k=3, bias=bias)), ("aux", dict(aux=aux, aux_mode=ops.MM_AUX_ADD, act=ops.MM_ACT_RELU))):
    C = torch.full((M + 8, N + 8), 7.0, device="cuda")
    be.mm(MMOp(A, K), MMOp(B, K), C, M, N, K, N + 8, ldaux=N, precision=1, **kw)
    torch.cuda.synchronize()
    pad = (C[:M, N:] != 7).nonzero(); tail = (C[M:] != 7).nonzero()
    r = ref + (bias.double() if "bias" in kw else 0)
    if name == "aux": r = torch.relu(ref) + aux.double()
    err = ((C[:M, :N].double() - r).norm() / r.norm()).item()
    print(name, "err", err, "pad writes", pad.shape[0], pad[:5].tolist(), "tail writes", tail.shape[0], tail[:5].tolist())
C = torch.zeros(M, N, device="cuda")
be.mm(MMOp(A, K), MMOp(B, K), C, M, N, K, N, precision=1, split_k=3)
torch.cuda.synchronize()
P = [(A[:, s * 256:(s + 1) * 256].double() @ B[:, s * 256:(s + 1) * 256].double().t()).flatten() for s in range(3)]
X = torch.stack(P, 1)
sol = torch.linalg.lstsq(X, C.double().flatten()[:, None]).solution.flatten()
print("split-K result as a combination of the three K-slices:", sol.tolist(), "residual", ((X @ sol - C.double().flatten()).norm() / C.double().norm()).item())
print("rows with error:", ((C.double() - ref).abs().amax(1) > 1e-2).nonzero().flatten()[:20].tolist(), "cols:", ((C.double() - ref).abs().amax(0) > 1e-2).nonzero().flatten()[:20].tolist())
