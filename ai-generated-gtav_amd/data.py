"""Dataset step and frame I/O on either side of the hot path.

* `strip_to_clip` — what the reference's loaders do per sample on the CPU (web_dataset.py:41-57,105-107, hf_dataset.py:22-41):
  `transforms.Compose([ToTensor(), SplitImages(), Resize((360, 640))])` on a decoded 270 x 2400 strip of five frames, as ONE HIP kernel
  on the uploaded uint8 strip (gtav_strip_to_frames); `actions_to_one_hot` is the host-side 25-way one-hot (web_dataset.py:22-38).
* `read_prompt_frame` — generate.py:150-153: a jpg / png start frame -> (1, 1, 3, 360, 640) in [0, 1].  The file is decoded on the host
  with PIL (the image has no GPU JPEG decoder); /255 and the antialiased resize run on the GPU (gtav_resize_frames).
* `write_video` — generate.py:244-246 / train_dit.py:458-463 hand uint8 frames (T, H, W, 3) to torchvision.io.write_video (PyAV + an
  H.264 encoder).  Neither exists in this image, so the frames are written as Motion-JPEG in an AVI container (PIL encodes the
  frames; the RIFF container is written here), or as `.npy` / PNG files; with torchvision installed an `.mp4` path goes to it.
"""
from __future__ import annotations

import io
import os
import struct
from typing import Sequence

import torch

from . import lib as _lib
from .dummy_dataset import actions_to_one_hot  # noqa: F401  (re-exported: web_dataset.py:22-38)


def _as_u8_hwc(img) -> torch.Tensor:
    """PIL image / numpy array / tensor -> contiguous uint8 (H, W, 3) CPU tensor."""
    if isinstance(img, torch.Tensor):
        t = img
    else:
        import numpy as np
        if hasattr(img, "convert"):
            img = img.convert("RGB")
        t = torch.from_numpy(np.array(img, copy=True))
    if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
        raise ValueError(f"expected a uint8 (H, W, 3) image, got {tuple(t.shape)} {t.dtype}")
    return t.contiguous()


@torch.inference_mode()
def strip_to_clip(strip, n_frames: int = 5, size: Sequence[int] = (360, 640), device=None) -> torch.Tensor:
    """Decoded strip image (H, n_frames * W, 3) uint8 -> clip (n_frames, 3, size[0], size[1]) float32 in [0, 1] on the GPU."""
    t = _as_u8_hwc(strip)
    H, Wt = t.shape[:2]
    if Wt % n_frames:
        raise ValueError(f"strip width {Wt} is not a multiple of {n_frames} frames")
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    td = t.to(dev)
    out = torch.empty((n_frames, 3, int(size[0]), int(size[1])), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().gtav_strip_to_frames(td.data_ptr(), H, Wt // n_frames, n_frames, out.data_ptr(), int(size[0]), int(size[1]),
                                                    _lib.current_stream()))
    return out


@torch.inference_mode()
def resize_frames(frames: torch.Tensor, size: Sequence[int] = (360, 640)) -> torch.Tensor:
    """(N, 3, H, W) float frames on the GPU -> (N, 3, size) with the antialiased bilinear filter of torchvision's tensor Resize."""
    x = frames.to(torch.float32).contiguous()
    assert x.is_cuda and x.dim() == 4 and x.shape[1] == 3
    out = torch.empty((x.shape[0], 3, int(size[0]), int(size[1])), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().gtav_resize_frames(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[2], x.shape[3], int(size[0]), int(size[1]),
                                                  _lib.current_stream()))
    return out


@torch.inference_mode()
def read_prompt_frame(path: str, size: Sequence[int] = (360, 640), device=None) -> torch.Tensor:
    """generate.py:150-153: `read_image(path)` -> float / 255 -> Resize(size) -> (1, 1, 3, H, W)."""
    from PIL import Image
    t = _as_u8_hwc(Image.open(path))
    return strip_to_clip(t, n_frames=1, size=size, device=device)[None]


# ------------------------------------------------------------------------------------------------------------------------
# video out
# ------------------------------------------------------------------------------------------------------------------------
def _avi_mjpeg(path: str, frames: torch.Tensor, fps: int, quality: int) -> None:
    from PIL import Image
    T, H, W, _ = frames.shape
    jpgs = []
    for i in range(T):
        buf = io.BytesIO()
        Image.fromarray(frames[i].numpy(), "RGB").save(buf, format="JPEG", quality=quality)
        b = buf.getvalue()
        jpgs.append(b + (b"\x00" if len(b) & 1 else b""))
    def chunk(tag, data):
        return tag + struct.pack("<I", len(data)) + data + (b"\x00" if len(data) & 1 else b"")
    def lst(tag, data):
        return b"LIST" + struct.pack("<I", len(data) + 4) + tag + data
    maxb = max(len(j) for j in jpgs)
    avih = struct.pack("<14I", 1000000 // fps, maxb * fps, 0, 0x10, T, 0, 1, maxb, W, H, 0, 0, 0, 0)
    strh = b"vids" + b"MJPG" + struct.pack("<IHHIIIIIIII", 0, 0, 0, 0, 1, fps, 0, T, maxb, 0xFFFFFFFF, 0) + struct.pack("<4h", 0, 0, W, H)
    strf = struct.pack("<IiiHH4sIiiII", 40, W, H, 1, 24, b"MJPG", W * H * 3, 0, 0, 0, 0)
    hdrl = lst(b"hdrl", chunk(b"avih", avih) + lst(b"strl", chunk(b"strh", strh) + chunk(b"strf", strf)))
    movi_body, idx, off = b"", b"", 4
    for j in jpgs:
        movi_body += b"00dc" + struct.pack("<I", len(j)) + j
        idx += b"00dc" + struct.pack("<III", 0x10, off, len(j))
        off += 8 + len(j)
    riff_body = b"AVI " + hdrl + lst(b"movi", movi_body) + chunk(b"idx1", idx)
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(riff_body)) + riff_body)


def write_video(path: str, frames: torch.Tensor, fps: int = 10, quality: int = 95) -> str:
    """uint8 frames (T, H, W, 3) -> file.  `.avi`: Motion-JPEG (always available); `.npy`: raw array; a directory: PNG files;
    `.mp4`: torchvision.io.write_video when torchvision + PyAV are installed, else the same frames as `<stem>.avi` (returned path)."""
    fr = frames.detach().to("cpu")
    if fr.dtype != torch.uint8 or fr.dim() != 4 or fr.shape[-1] != 3:
        raise ValueError(f"expected uint8 (T, H, W, 3) frames, got {tuple(fr.shape)} {fr.dtype}")
    fr = fr.contiguous()
    ext = os.path.splitext(path)[1].lower()
    if ext == ".mp4":
        try:
            from torchvision.io import write_video as tv_write   # generate.py:245
            tv_write(path, fr, fps=fps)
            return path
        except Exception:
            path, ext = os.path.splitext(path)[0] + ".avi", ".avi"
    if ext == ".npy":
        import numpy as np
        np.save(path, fr.numpy())
    elif ext == ".avi":
        _avi_mjpeg(path, fr, fps, quality)
    else:
        from PIL import Image
        os.makedirs(path, exist_ok=True)
        for i in range(fr.shape[0]):
            Image.fromarray(fr[i].numpy(), "RGB").save(os.path.join(path, f"frame_{i:04d}.png"))
    return path


def read_avi_mjpeg(path: str) -> torch.Tensor:
    """Inverse of the `.avi` writer (tests, round trips): uint8 (T, H, W, 3)."""
    import numpy as np
    from PIL import Image
    data = open(path, "rb").read()
    assert data[:4] == b"RIFF" and data[8:12] == b"AVI "
    out, pos = [], data.find(b"movi") + 4
    while data[pos:pos + 4] == b"00dc":
        n = struct.unpack("<I", data[pos + 4:pos + 8])[0]
        out.append(torch.from_numpy(np.asarray(Image.open(io.BytesIO(data[pos + 8:pos + 8 + n])).convert("RGB")).copy()))
        pos += 8 + n + (n & 1)
    return torch.stack(out)
