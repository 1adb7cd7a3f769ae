"""synthetic COCO-format dataset shared by the golden generator and the tests (no reference code)"""
import json
import os

import numpy as np


def synthetic_coco(root, n_img=7, seed=3, classes=('echinus', 'starfish', 'holothurian', 'scallop')):
    """tiny COCO-format dataset on disk: .npy BGR uint8 images + annotation json with the corner
    cases `_parse_ann_info` / `_filter_imgs` handle (crowd, ignore, zero-area, out-of-image,
    foreign category, an image without annotations, a too-small image)"""
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, 'imgs'), exist_ok=True)
    images, anns = [], []
    aid = 1
    sizes = [(96, 128), (128, 96), (80, 120), (100, 100), (90, 150), (20, 200), (64, 72)]
    for i in range(n_img):
        h, w = sizes[i % len(sizes)]
        name = f'{i:03d}.npy'
        np.save(os.path.join(root, 'imgs', name), rng.randint(0, 256, (h, w, 3), dtype=np.uint8))
        images.append(dict(id=100 + i, file_name=name, height=h, width=w))
        if i == 4:
            continue                      # image without annotations
        for j in range(rng.randint(1, 5)):
            bw, bh = rng.uniform(8, w * 0.6), rng.uniform(8, h * 0.6)
            x, y = rng.uniform(0, w - bw), rng.uniform(0, h - bh)
            anns.append(dict(id=aid, image_id=100 + i, category_id=int(rng.randint(1, 5)),
                             bbox=[float(x), float(y), float(bw), float(bh)], area=float(bw * bh), iscrowd=0))
            aid += 1
    extra = [dict(image_id=100, category_id=1, bbox=[5., 5., 30., 20.], area=600., iscrowd=1),
             dict(image_id=100, category_id=2, bbox=[5., 5., 30., 20.], area=600., iscrowd=0, ignore=True),
             dict(image_id=101, category_id=2, bbox=[10., 10., 0.5, 20.], area=10., iscrowd=0),
             dict(image_id=101, category_id=3, bbox=[-50., -50., 20., 20.], area=400., iscrowd=0),
             dict(image_id=102, category_id=9, bbox=[10., 10., 20., 20.], area=400., iscrowd=0),
             dict(image_id=102, category_id=1, bbox=[10., 10., 20., 20.], area=0., iscrowd=0)]
    for e in extra:
        anns.append(dict(e, id=aid))
        aid += 1
    cats = [dict(id=k + 1, name=c) for k, c in enumerate(classes)] + [dict(id=9, name='other')]
    ann_file = os.path.join(root, 'ann.json')
    json.dump(dict(images=images, annotations=anns, categories=cats), open(ann_file, 'w'))
    return ann_file, os.path.join(root, 'imgs')


def synthetic_voc(root, n_img=6, seed=4):
    """tiny VOC2007-style tree: JPEGImages/<id>.jpg.npy is NOT used (annotations only carry sizes),
    Annotations/<id>.xml with difficult / foreign-class / tiny objects, ImageSets/Main/test.txt"""
    rng = np.random.RandomState(seed)
    base = os.path.join(root, 'VOC2007')
    os.makedirs(os.path.join(base, 'Annotations'), exist_ok=True)
    os.makedirs(os.path.join(base, 'ImageSets', 'Main'), exist_ok=True)
    names = ['aeroplane', 'bicycle', 'bird', 'person', 'dog', 'unicorn']
    ids = []
    for i in range(n_img):
        img_id = f'{i:06d}'
        ids.append(img_id)
        w, h = int(rng.randint(100, 300)), int(rng.randint(100, 300))
        objs = []
        for j in range(rng.randint(0 if i == 3 else 1, 5)):
            bw, bh = rng.randint(4, w // 2), rng.randint(4, h // 2)
            x, y = rng.randint(1, w - bw), rng.randint(1, h - bh)
            objs.append(f'<object><name>{names[rng.randint(len(names))]}</name>'
                        f'<difficult>{int(rng.rand() < 0.25)}</difficult><bndbox><xmin>{x}</xmin><ymin>{y}</ymin>'
                        f'<xmax>{x + bw}</xmax><ymax>{y + bh}.0</ymax></bndbox></object>')
        size = f'<size><width>{w}</width><height>{h}</height><depth>3</depth></size>' if i != 2 else \
            f'<size><width>{w}</width><height>20</height><depth>3</depth></size>'
        open(os.path.join(base, 'Annotations', img_id + '.xml'), 'w').write(
            f'<annotation><filename>{img_id}.jpg</filename>{size}{"".join(objs)}</annotation>')
    lst = os.path.join(base, 'ImageSets', 'Main', 'test.txt')
    open(lst, 'w').write('\n'.join(ids) + '\n')
    return lst, base + '/'
