"""Shared test helpers (tests only)."""
import numpy as np


def flat_tree_preorder(nodes, leaves, items, root):
    """Flattened tree (inner nodes with child boxes + leaves) -> the oracle's DFS pre-order dump:
    (leaf flags, member counts, boxes[n,6] with NaN for the root's box, items in leaf order)."""
    leaf_flags, counts, boxes, out_items = [], [], [], []
    stack = [(root, None)]
    while stack:
        link, box = stack.pop()
        boxes.append(box if box is not None else [np.nan] * 6)
        if link < 0:
            lf = leaves[-1 - link]
            leaf_flags.append(1)
            counts.append(int(lf["count"]))
            out_items.extend(items[int(lf["first"]):int(lf["first"]) + int(lf["count"])].tolist())
        else:
            n = nodes[link]
            leaf_flags.append(0)
            counts.append(0)
            stack.append((int(n["right"]), list(n["rmin"]) + list(n["rmax"])))
            stack.append((int(n["left"]), list(n["lmin"]) + list(n["lmax"])))
    return (np.array(leaf_flags, np.int32), np.array(counts, np.int32), np.array(boxes, np.float32),
            np.array(out_items, np.int64))


def psnr(a, b, peak=1.0):
    a = np.clip(np.asarray(a, np.float64), 0, peak)
    b = np.clip(np.asarray(b, np.float64), 0, peak)
    mse = np.mean((a - b) ** 2)
    return 99.0 if mse == 0 else 10 * np.log10(peak * peak / mse)
