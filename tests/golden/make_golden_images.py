"""Generates the image-decode fixture F14: two small image FILES (a JPEG and a PNG of the same synthetic picture) and the resized uint8
arrays `learning_embeddings_amd.image_store.decode_u8` must produce from them.

What it pins: the decode -> resize half of the reference's input pipeline (oe_h.py:668-677, 700-712, 1463-1471: cv2.imread ->
ToPILImage -> Resize((224, 224)) -> ToTensor), as far as this image allows.  cv2 and torchvision are NOT installed here, so the
expected arrays come from PIL (Pillow 12.2, libjpeg-turbo) -- the library the reference's ToPILImage / Resize call into for the resize;
the JPEG *decode* of the reference is cv2's (its own libjpeg build), whose IDCT / chroma upsampling may differ from PIL's by 1-2 / 255 on
JPEG files (PNG is lossless: identical).  That gap is stated in DESIGN.md; this fixture keeps OUR decode from drifting.

Run from the repo root:  python tests/golden/make_golden_images.py"""
import os
import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def picture(h=150, w=200, seed=7):
    r = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 255 // (w - 1)), (yy * 255 // (h - 1)), ((xx + yy) * 255 // (h + w - 2))], axis=2).astype(np.int32)
    img[40:90, 60:140] = [250, 20, 60]                                 # a saturated block: chroma edges are where JPEG decoders differ
    img += r.randint(-12, 13, size=img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)                       # R, G, B


if __name__ == '__main__':
    rgb = picture()
    d = os.path.join(HERE, 'images')
    os.makedirs(d, exist_ok=True)
    Image.fromarray(rgb).save(os.path.join(d, 'picture.png'))
    Image.fromarray(rgb).save(os.path.join(d, 'picture.jpg'), quality=90)
    out = {'rgb': rgb}
    for ext in ('png', 'jpg'):
        with Image.open(os.path.join(d, 'picture.' + ext)) as im:
            full = np.asarray(im.convert('RGB'))
            res = np.asarray(im.convert('RGB').resize((224, 224), Image.BILINEAR))
        out['decoded_rgb_' + ext] = full                               # the decoder's output before the resize
        out['resized_bgr_' + ext] = np.ascontiguousarray(res[:, :, ::-1])   # what decode_u8 returns (cv2's B, G, R order)
    np.savez_compressed(os.path.join(HERE, 'F14_image_decode.npz'), **out)
    print({k: v.shape for k, v in out.items()})
