"""What a reduced-precision recurrent state does to causal linear attention (BASELINE configs[4] names an "fp8
KV-state").  CPU experiment, no product code: the recurrent form  S += phi(k) (x) v ;  z += phi(k) ;
out = phi(q).S / (phi(q).z + eps)  with S and z ROUNDED TO THE STORAGE TYPE AFTER EVERY TOKEN (that is what keeping the
state in that type in HBM means), against the f64 recurrence.  Storage types: f32, bf16, fp8 e4m3 with a per-row
power-of-two scale refreshed every token (the most favourable fp8 layout).

    python tools/kv_state_precision.py            -> table on stdout (committed as profiles/r02_kv_state_precision.txt)
"""
import torch


def phi(x):
    return torch.nn.functional.elu(x) + 1


def q_f32(x):
    return x.float().double()


def q_bf16(x):
    return x.float().bfloat16().double()


def q_fp8_rowscaled(x):
    """e4m3 with one power-of-two scale per state row (max |row| mapped just under 448)."""
    a = x.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
    s = torch.pow(2.0, torch.floor(torch.log2(448.0 / a)))
    return ((x * s).float().to(torch.float8_e4m3fn).double()) / s


def run(T, quant, seed=0, H=4, D=64):
    g = torch.Generator().manual_seed(seed)
    q, k, v = (torch.randn(T, H, D, generator=g, dtype=torch.float64) for _ in range(3))
    qf, kf = phi(q), phi(k)
    S = torch.zeros(H, D, D, dtype=torch.float64)
    z = torch.zeros(H, D, dtype=torch.float64)
    Sr, zr = S.clone(), z.clone()
    errs = []
    for t in range(T):
        upd = kf[t].unsqueeze(-1) * v[t].unsqueeze(-2)
        Sr = Sr + upd
        zr = zr + kf[t]
        S = quant(S + upd)
        z = quant((z + kf[t]).unsqueeze(-2)).squeeze(-2)
        ref = torch.einsum("hd,hdm->hm", qf[t], Sr) / ((qf[t] * zr).sum(-1, keepdim=True) + 1e-6)
        out = torch.einsum("hd,hdm->hm", qf[t], S) / ((qf[t] * z).sum(-1, keepdim=True) + 1e-6)
        errs.append(((out - ref).abs().max() / ref.abs().max().clamp_min(1e-9)).item())
    e = torch.tensor(errs)
    return e[: T // 8].max().item(), e[T // 2:].max().item(), e[-T // 8:].mean().item()


def main():
    print("relative output error of recurrent linear attention vs f64, state rounded to the storage type every token")
    print("(max over the first eighth of the tokens | max over the second half | mean over the last eighth)")
    for T in (1024, 4096):
        for name, fn in (("f32", q_f32), ("bf16", q_bf16), ("fp8 e4m3, per-row scale", q_fp8_rowscaled)):
            a, b, c = run(T, fn)
            print("T=%5d  state %-24s  %.2e | %.2e | %.2e" % (T, name, a, b, c))
    print("bytes of state traffic per generated token per song (12 layers x 8 heads x (64 x 64 + 64), read + write):")
    for name, s in (("f32", 4), ("bf16", 2), ("fp8", 1)):
        print("  %-5s %.2f MB" % (name, 12 * 8 * (64 * 64 + 64) * s * 2 / 1e6))
    print("weights read per token (all songs of a step share them): 156 MB f32")


if __name__ == "__main__":
    main()
