import numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
from epic_amd import epic_harmonic as eh
E = eh._epic
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
def run(x, which):
    d_in = torch.from_numpy(x).to(dev); d_out = torch.empty_like(d_in)
    assert E.epic_hip_eval_math(d_in.data_ptr(), d_out.data_ptr(), x.size, which, s) == 0
    torch.cuda.synchronize(); return d_out.cpu().numpy()
x = np.arange(np.float32(1.0).view(np.uint32), np.float32(6.0).view(np.uint32)+1, 64, dtype=np.uint32).view(np.float32)
got = run(x, 3); ref = np.log(x.astype(np.float64))
err = (got - ref) / np.spacing(np.maximum(np.abs(ref), 1e-3).astype(np.float32))
print("df_ln: max ulp (capped at |ln|>=1e-3)", np.abs(err).max(), "mean", err.mean(), "worst x", x[np.abs(err).argmax()], got[np.abs(err).argmax()], ref[np.abs(err).argmax()])
xe = -np.abs(np.random.default_rng(0).standard_cauchy(2_000_000)).astype(np.float32).clip(0, 120)
xe[:7] = [0, -1e-9, -103.9, -104, -200, -1e6, -3.4e38]
got = run(xe, 2); ref = np.exp(xe.astype(np.float64))
nrm = ref > 1e-37
err = (got[nrm] - ref[nrm]) / np.spacing(ref[nrm].astype(np.float32))
print("df_exp: max ulp", np.abs(err).max(), "mean", err.mean(), "worst x", xe[nrm][np.abs(err).argmax()], "first7", got[:7])
print("nan?", np.isnan(got).any(), "inf?", np.isinf(got).any())
xs = np.array([0, -1e-9, -0.25, -1.0, -50.0, -103.9, -104, -200, -1e6], dtype=np.float32)
print("exp .y:", run(xs, 4), " ref", np.exp(xs.astype(np.float64)))
ls = np.array([1.0, 1.5, 2.0, 3.999, 4.0, 5.9], dtype=np.float32)
print("ln .y:", run(ls, 5), " ref", np.log(ls.astype(np.float64)))
