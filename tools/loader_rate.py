"""Developer tool (GPU box): how many images per second the batch producer delivers.
Synthetic JPEGs of photo size (500 x 375) on local disk; compared: (a) the reference's path — PIL decode + PIL resize + float
conversion in the DataLoader workers, (b) this repo's — PIL decode only in the workers, raw uint8 copied to the GPU (pinned, side
stream), zsg_resize_u8 + zsg_u8hwc_to_nhwc4 there.
usage: python tools/loader_rate.py [n_images] [workers ...]"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zsgnet_pytorch_amd import dat_loader as D
from zsgnet_pytorch_amd._lib import lib, check, stream_ptr


class Raw(torch.utils.data.Dataset):
    def __init__(self, files, resize):
        self.files, self.resize = files, resize

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        import PIL.Image
        img = PIL.Image.open(self.files[i]).convert("RGB")
        if self.resize:                                             # the reference's item: resize + float conversion on the host
            img = img.resize((300, 300))
            return torch.from_numpy(np.asarray(img).transpose(2, 0, 1).astype(np.float64)).float().div_(255)
        return torch.from_numpy(np.asarray(img).copy())             # raw uint8 [h, w, 3]


def main():
    import PIL.Image
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    workers = [int(a) for a in sys.argv[2:]] or [1, 4, 16]
    rng = np.random.default_rng(0)
    td = tempfile.mkdtemp()
    files = []
    for i in range(64):
        h, w = (375, 500) if i % 3 else (500, 333)
        yy, xx = np.mgrid[0:h, 0:w]
        a = np.clip(127 + 90 * np.sin(yy / 17.0 + i)[..., None] * np.cos(xx / 23.0)[..., None] * np.array([1, .6, -.7]) + rng.normal(0, 12, (h, w, 3)), 0, 255)
        f = os.path.join(td, f"{i}.jpg")
        PIL.Image.fromarray(a.astype(np.uint8)).save(f, quality=90)
        files.append(f)
    files = (files * ((n + 63) // 64))[:n]
    print(f"{n} JPEGs (500x375 / 333x500), batch 16 -> 300x300; images per second")
    for nw in workers:
        for mode in ("host resize (reference path)", "GPU resize (zsg_resize_u8)"):
            host = mode.startswith("host")
            ds = Raw(files, resize=host)
            dl = torch.utils.data.DataLoader(ds, batch_size=16, num_workers=nw, pin_memory=True, persistent_workers=False,
                                             collate_fn=(None if host else (lambda b: dict(zip(("img", "img_hw"), D.flatten_raw(b))))))
            it = dl if host else D.DevicePrefetcher(dl, "cuda", resize_hw=(300, 300))      # (side-stream copies + resize, as the trainer's loader)
            t0, cnt = None, 0
            for bi, b in enumerate(it):
                if bi == 2:
                    torch.cuda.synchronize()
                    t0, cnt = time.perf_counter(), 0
                if host:
                    x = b.cuda(non_blocking=True)
                    cnt += x.shape[0]
                else:
                    u8 = b["img"]
                    out = torch.empty(u8.shape[0], 300, 300, 4, device="cuda")
                    check(lib.zsg_u8hwc_to_nhwc4(u8.data_ptr(), u8.shape[0] * 300 * 300, out.data_ptr(), stream_ptr()), "u8")
                    cnt += u8.shape[0]
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"  workers {nw:2d}  {mode:32s} {cnt / dt:8.1f} img/s  ({cnt / dt / nw:7.1f} per worker)")


if __name__ == "__main__":
    main()
