"""Background half of test_gpu_fullshape.py::test_learnable_task_reaches_the_same_accuracy (started by tests/conftest.py): the CPU
oracle (torch-CPU fp32 restatement of the reference, oracle/zsg_oracle.py) trained on O.learnable_batch from the same start, the same
batches and the same LSTM start states as the HIP model in the test.  Writes the per-step losses and the trained state.
usage: python tests/oracle_learn_worker.py <S> <steps> <out.pt>"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import zsg_oracle as O  # noqa: E402


def main():
    S, steps, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    ratios, scales = O.default_ratios_scales()
    B, lr_ = 16, 1e-3
    sd = O.seeded_state_dict("resnet50", 3)
    params = {k: v.clone().requires_grad_() for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    buffers = {k: v.clone() for k, v in sd.items() if k not in params}
    opt_ref = torch.optim.Adam(list(params.values()), lr=lr_, betas=(0.9, 0.99))
    anc = torch.from_numpy(O.create_anchors(O.feat_sizes_for(S, S), ratios, scales).astype(np.float32))
    gq = torch.Generator().manual_seed(8)
    losses = []
    for it in range(steps):
        bt = O.learnable_batch(B, S, seed=100 + it)
        h0, c0 = torch.randn(2, B, 128, generator=gq), torch.randn(2, B, 128, generator=gq)
        lr, _ = O.cpu_train_step(params, buffers, opt_ref, bt, h0, c0, anc, arch="resnet50")
        losses.append(float(lr["loss"].detach()))
    sd_ref = {k: v.detach() for k, v in params.items()}
    sd_ref.update(buffers)
    torch.save({"losses": losses, "sd": sd_ref}, out + ".tmp")
    os.replace(out + ".tmp", out)


if __name__ == "__main__":
    main()
