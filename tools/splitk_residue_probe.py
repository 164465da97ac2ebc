"""Developer probe: does any sdt_gemm_nt_bf16 launch of a train_step leave its split-K workspace non-zero (include/sdt.h contract)?"""
import sys

import torch

sys.path.insert(0, ".")
from tests.helpers import build_hip_states, make_case, to_dev  # noqa: E402
from stable_diffusion_training_amd import ops  # noqa: E402
from stable_diffusion_training_amd import training_utils as tu  # noqa: E402

dev = torch.device("cuda:0")
size = sys.argv[1] if len(sys.argv) > 1 else "tiny"
case = make_case(size, B=2, image=64)
orig = ops._gemm_nt
seen = set()


def checked(A, Bt, out, M, N, Kc, taps, lda, ldb, bts, bias, rowbias, residual, rpb, mode, geom, gn_stats=None, gn_groups=0):
    orig(A, Bt, out, M, N, Kc, taps, lda, ldb, bts, bias, rowbias, residual, rpb, mode, geom, gn_stats, gn_groups)
    ws = ops._SPLITK_WS.get(out.device)
    if ws is not None:
        torch.cuda.synchronize()
        nz = int(torch.count_nonzero(ws))
        if nz:
            key = (M, N, Kc, taps, mode)
            if key not in seen:
                seen.add(key)
                idx = torch.nonzero(ws.view(torch.int32)).flatten()
                print(f"residue after gemm_nt M={M} N={N} Kc={Kc} taps={taps} mode={mode} bias={bias is not None} rowbias={rowbias is not None} "
                      f"res={residual is not None} gn={gn_stats is not None}: {nz} bytes non-zero, int32 index range [{int(idx.min())}, {int(idx.max())}]", flush=True)
            ws.zero_()


ops._gemm_nt = checked
tc, (us, ts, ue, te, vae, sc, _) = build_hip_states(case, dev)
tu.train_step(us, ts, None, None, to_dev(case["batch"], dev), torch.Generator(device=dev), vae, sc,
              strip_bos_eos_token=False, rand=to_dev(case["rand"], dev))
torch.cuda.synchronize()
print("done;", len(seen), "offending shapes")
