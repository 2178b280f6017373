"""Container-only: golden vectors for the training-time augmentation (tests/golden/augment.npz).

The reference's data package cannot be imported here (nibabel / skimage / torchvision are not installed), so the pieces that are pure
numpy / scipy / PIL are compiled from the files WHERE THEY LIE with `ast` (function / class definitions only; nothing is copied
into the repo) and run on seeded inputs with their random draws pinned:
  * PaddingCenterCrop, RandomHorizontallyFlip, RandomVerticallyFlip .. /root/reference/data/augmentations.py:223-264, 308-331 (PIL)
  * augment_gamma ..................................................... /root/reference/data/ac17_dataloader.py:22-57 (numpy)
  * AC17_2DLoad.random_elastic_deformation ............................. /root/reference/data/ac17_dataloader.py:260-287 (scipy)
RandomRotate (augmentations.py:392-412) goes through torchvision.transforms.functional.affine, which is not installed: it stays
an unpinned restatement (tests/test_augment.py holds its known-answer tests).
"""
import ast
import os
import sys

import numpy as np

REF = "/root/reference/data"


def extract(path, names, ns):
    tree = ast.parse(open(path).read())
    keep = []
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            keep.append(node)
    got = {n.name for n in keep}
    assert got == set(names), (got, names)
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return ns


def main():
    sys.dont_write_bytecode = True
    import numbers, random
    from PIL import Image, ImageOps
    from scipy.ndimage import gaussian_filter, map_coordinates
    out = {}
    # ---- crop / pad / flips (PIL)
    ns = dict(np=np, Image=Image, ImageOps=ImageOps, numbers=numbers, random=random, object=object)
    extract(os.path.join(REF, "augmentations.py"), ["PaddingCenterCrop", "RandomHorizontallyFlip", "RandomVerticallyFlip"], ns)
    crop = ns["PaddingCenterCrop"](64)
    cases = [(75, 70), (65, 76), (50, 45), (51, 63), (75, 50), (49, 78), (64, 64), (63, 65)]
    for i, (h, w) in enumerate(cases):
        r = np.random.default_rng(40 + i)
        img = r.integers(0, 2000, size=(h, w)).astype(np.uint32)
        seg = r.integers(0, 4, size=(h, w)).astype(np.uint8)
        pi, ps = Image.fromarray(img.astype(np.int32), mode="I"), Image.fromarray(seg, mode="L")
        ci, cs = crop(pi, ps)
        hf, vf = bool(i & 1), bool(i & 2)
        if hf:
            ci, cs = ns["RandomHorizontallyFlip"](p=2.0)(ci, cs)      # p > 1: random.random() < p always
        if vf:
            ci, cs = ns["RandomVerticallyFlip"](p=2.0)(ci, cs)
        out["crop%d.img" % i] = img.astype(np.uint16); out["crop%d.seg" % i] = seg
        out["crop%d.flags" % i] = np.array([hf, vf])
        out["crop%d.out_img" % i] = np.array(ci).astype(np.uint16); out["crop%d.out_seg" % i] = np.array(cs, dtype=np.uint8)
    # ---- gamma (numpy); the two uniform draws are pinned by seeding numpy's global generator right before the call
    ns = dict(np=np)
    extract(os.path.join(REF, "ac17_dataloader.py"), ["augment_gamma"], ns)
    for i in range(4):
        r = np.random.default_rng(70 + i)
        x = r.integers(0, 1500, size=(64, 48)).astype(np.float64)
        np.random.seed(500 + i)
        coin = np.random.random()
        gamma = np.random.uniform(0.5, 1) if coin < 0.5 else np.random.uniform(1, 2)
        np.random.seed(500 + i)
        y = ns["augment_gamma"](x.copy())
        out["gamma%d.x" % i] = x; out["gamma%d.gamma" % i] = np.array(gamma); out["gamma%d.y" % i] = y
    # ---- elastic deformation (scipy): same RandomState stream -> the two uniform fields are reproducible
    ns = dict(np=np, gaussian_filter=gaussian_filter, map_coordinates=map_coordinates)
    tree = ast.parse(open(os.path.join(REF, "ac17_dataloader.py")).read())
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "AC17_2DLoad"][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "random_elastic_deformation"]
    exec(compile(ast.Module(body=fn, type_ignores=[]), "ac17_dataloader.py", "exec"), ns)
    deform = ns["random_elastic_deformation"]
    for i, (h, w) in enumerate([(64, 64), (48, 80)]):
        r = np.random.default_rng(90 + i)
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.sin(yy / 7.0) * 50 + np.cos(xx / 5.0) * 30 + r.standard_normal((h, w))
        seg = ((yy - h / 2) ** 2 + (xx - w / 2) ** 2 < (h / 4) ** 2).astype(np.float64) * 2 + (yy > h * 0.7).astype(np.float64)
        stacked = np.stack([img, seg], 2)
        rs = np.random.RandomState(1234 + i)
        u1 = rs.rand(h, w); u2 = rs.rand(h, w)
        res = deform(None, stacked, alpha=500, sigma=20, random_state=np.random.RandomState(1234 + i))
        f32 = np.float32
        out["deform%d.in" % i] = stacked.astype(f32); out["deform%d.u1" % i] = u1.astype(f32); out["deform%d.u2" % i] = u2.astype(f32)
        out["deform%d.out" % i] = res.astype(f32)
        out["deform%d.dx" % i] = (gaussian_filter(2 * u1 - 1, 20, mode="constant", cval=0) * 500).astype(f32)
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "augment.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, {k.split(".")[0].rstrip("0123456789") for k in out})


if __name__ == "__main__":
    main()
