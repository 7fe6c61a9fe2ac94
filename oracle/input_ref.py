"""Test infrastructure (only tests/ may import it): CPU restatement of the reference's per-sample input preparation.

Follows /root/reference data/realestate10k_dataio.py:333-441 for ONE sample whose frame ids and ray selection are
given: centre square crop (utils_training/data_util.py:116-121), `rgb.astype(np.float32) / 127.5 - 1` (:353, :437),
query colours at the selected pixels (:385-390), intrinsics un-normalised by the frame size with the principal point
divided by the crop scale (:51-55, :343-348).  PINNED by tests/golden/input.npz: one sample produced by the reference's
own `RealEstate10k.__getitem__` from a synthetic scene (tests/golden/make_golden_input.py), reproduced bit for bit in
tests/test_input_golden.py.  Not restated: the `cv2.resize` of 360-row frames (:342-343; cv2 is not in this image and
the shard format stores frames at 256x455)."""
import numpy as np


def square_crop(img):
    m = np.amin(img.shape[:2])
    c = np.array(img.shape[:2]) // 2
    return img[c[0] - m // 2:c[0] + m // 2, c[1] - m // 2:c[1] + m // 2]


def intrinsics_4x4(intr_norm, Hs, Ws):
    fx, fy, cx, cy = intr_norm
    K = np.array([[fx, 0, cx, 0], [0, fy, cy, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)
    K[0] *= Ws
    K[1] *= Hs
    K[0, 2] = K[0, 2] / (Ws / min(Hs, Ws))
    K[1, 2] = K[1, 2] / (Hs / min(Hs, Ws))
    return K.astype(np.float32)


def prepare_sample(frames_u8, c2w, intr_norm, ids, ray_pix):
    """frames (N,Hs,Ws,3) uint8; ids = (ctx0, ctx1, query); ray_pix linear pixel ids in the cropped query frame."""
    Hs, Ws = frames_u8.shape[1:3]
    ctx = np.stack([square_crop(frames_u8[i]).astype(np.float32) / 127.5 - 1 for i in ids[:2]])
    q = square_crop(frames_u8[ids[2]]).astype(np.float32) / 127.5 - 1
    S = q.shape[0]
    uv = np.stack((ray_pix % S, ray_pix // S), -1).astype(np.float32)
    return {"context": {"rgb": ctx, "cam2world": np.stack([c2w[i] for i in ids[:2]]).astype(np.float32),
                        "intrinsics": np.stack([intrinsics_4x4(intr_norm[i], Hs, Ws) for i in ids[:2]])},
            "query": {"rgb": q.reshape(-1, 3)[ray_pix][None], "cam2world": c2w[ids[2]][None].astype(np.float32),
                      "intrinsics": intrinsics_4x4(intr_norm[ids[2]], Hs, Ws)[None], "uv": uv[None]}}
