"""Tokenisers of the reference's dataloaders: ``infos/<name>.json`` -> the batch
contract the model consumes.  Drop-in for ``plankassembly/datasets`` (same class names, constructor arguments and
method names), numpy only - shapely / PythonOCC are not needed to READ the prepared info files.

* ``quantize_values`` / ``dequantize_values``            reference plankassembly/datasets/data_utils.py:6-21
* ``LineDataset.prepare_input_sequence``                  reference line_data.py:34-83   (pinned: tests/golden/data_tokens.npz)
* ``LineDataset.prepare_output_sequence``                 reference line_data.py:85-109  (pinned)
* ``LineDataset.__getitem__``                             reference line_data.py:111-142 (info JSON schema: SURVEY appendix B)
* ``SidefaceDataset.prepare_input_sequence``              reference sideface_data.py:137-189 incl. the empty case (pinned)

Two things need computational geometry the image does not have and stay out of scope: the shapely line-noise
augmentation (data_utils.py:24-75) is restated here for the straight 2-point segments the info files hold
(``add_noise``; unpinned - there is no shapely to compare with), and side-face extraction from the line drawing
(sideface_data.py:21-135: polygonize + STRtree merge) is not reimplemented: ``SidefaceDataset`` reads the side-face
boxes from the info file when a preprocessing step has stored them (keys ``faces`` / ``faceviews``) and raises
otherwise.
"""
from __future__ import annotations

import json
import os

import numpy as np
import torch


def quantize_values(verts, n_bits=9):
    """[-1, 1] floats -> integers in [0, 2**n_bits - 1] (truncation, as the reference's ``astype('long')``)."""
    rq = 2 ** n_bits - 1
    return ((np.asarray(verts) - (-1)) * rq / 2).astype("long")


def dequantize_values(quantized_verts, n_bits=9):
    rq = 2 ** n_bits - 1
    return (np.asarray(quantized_verts) * 2 / rq + (-1)).astype("float")


def parse_splits_list(splits):
    """reference dataset/data_utils.py:28-46: '.json' entries are info files, '.txt' entries list one per line."""
    if isinstance(splits, str):
        splits = splits.split()
    info_files = []
    for split in splits:
        ext = os.path.splitext(split)[1]
        if ext == ".json":
            info_files.append(split)
        elif ext == ".txt":
            with open(split) as f:
                info_files += [ln.rstrip() for ln in f]
        else:
            raise NotImplementedError("%s not a valid info_file type" % split)
    return info_files


def _segment_points(svg):
    """GeoJSON LineString (string or dict) -> float [n, 2] vertices."""
    g = json.loads(svg) if isinstance(svg, str) else svg
    return np.asarray(g["coordinates"], dtype=float).reshape(-1, 2)


def _interpolate(pts, dist):
    """Point at arc length ``dist`` along the polyline (negative: from the end), clamped - shapely's
    line_interpolate_point."""
    seg = np.linalg.norm(np.diff(pts, axis=0), axis=1)
    total = float(seg.sum())
    if dist < 0:
        dist = total + dist
    dist = min(max(dist, 0.0), total)
    acc = 0.0
    for i, s in enumerate(seg):
        if dist <= acc + s or i == len(seg) - 1:
            t = 0.0 if s == 0 else (dist - acc) / s
            return pts[i] + t * (pts[i + 1] - pts[i])
        acc += s
    return pts[-1]


def add_noise(lines, views, types, noise_ratio, noise_length, rng=np.random):
    """The reference's drawing noise (data_utils.py:24-75) on vertex arrays instead of shapely geometries: a random
    subset of the lines is deleted or shortened from one end by up to ``noise_length``; same sequence of draws from
    numpy's global generator."""
    lines = list(lines)
    num_select = rng.randint(low=1, high=np.ceil(len(lines) * noise_ratio) + 1)
    indices = rng.choice(len(lines), num_select, replace=False)
    for index in indices:
        if rng.random() > 0.5:
            lines[index] = None
            continue
        pts = lines[index]
        length = float(np.linalg.norm(np.diff(pts, axis=0), axis=1).sum())
        noise = np.round(rng.random() * noise_length, 3)
        if length <= noise:
            lines[index] = None
        elif rng.random() > 0.5:
            lines[index] = np.stack([_interpolate(pts, 0.0), _interpolate(pts, -noise)])
        else:
            lines[index] = np.stack([_interpolate(pts, noise), _interpolate(pts, length)])
    keep = [i for i, ln in enumerate(lines) if ln is not None]
    return [lines[i] for i in keep], [views[i] for i in keep], [types[i] for i in keep]


def _bounds(lines):
    """[xmin, ymin, xmax, ymax] per vertex array (shapely.bounds)."""
    return np.array([[p[:, 0].min(), p[:, 1].min(), p[:, 0].max(), p[:, 1].max()] for p in lines], dtype=float).reshape(-1, 4)


class _TokenisedDataset(torch.utils.data.Dataset):
    def __init__(self, root, info_files, token, cfg, augmentation=False):
        self.root = root
        self.info_files = info_files
        self.augmentation = augmentation
        self.token = token
        self.vocab_size = cfg.VOCAB_SIZE
        self.num_input_dof = cfg.NUM_INPUT_DOF
        self.max_input_length = cfg.MAX_INPUT_LENGTH
        self.max_output_length = cfg.MAX_OUTPUT_LENGTH
        self.num_bits = cfg.NUM_BITS
        self.aug_ratio = cfg.get("AUG_RATIO", 0.0) if hasattr(cfg, "get") else getattr(cfg, "AUG_RATIO", 0.0)
        self.noise_ratio = cfg.get("NOISE_RATIO", 0.0) if hasattr(cfg, "get") else getattr(cfg, "NOISE_RATIO", 0.0)
        self.noise_length = cfg.get("NOISE_LENGTH", 0.0) if hasattr(cfg, "get") else getattr(cfg, "NOISE_LENGTH", 0.0)

    def __len__(self):
        return len(self.info_files)

    def _sorted_tokens(self, boxes, views, extra=None):
        """Shared by lines and side faces: quantise, sort by (view, xmin, xmax, ymin, ymax), position within the view,
        coordinate index, 4 tokens per primitive."""
        value = quantize_values(np.array(boxes), self.num_bits)
        view = np.array(views, dtype="long")
        with_view = np.concatenate((value, view[..., np.newaxis]), axis=1)
        order = np.lexsort(with_view.T[[3, 1, 2, 0, 4]])          # last key is primary: view, then column 0, 2, 1, 3
        value = value[order].flatten()
        view = view[order]
        extra = None if extra is None else np.array(extra)[order]
        _, counts = np.unique(view, return_counts=True)
        pos = np.concatenate([np.arange(c) for c in counts])
        coord = np.arange(len(value)) % self.num_input_dof
        return value, np.repeat(pos, 4), coord, np.repeat(view, 4), None if extra is None else np.repeat(extra, 4)

    def _pad_inputs(self, value, ids):
        """END, then PAD up to MAX_INPUT_LENGTH - 1 value tokens; id rows are zero at END / PAD (line_data.py:58-72:
        the value row gets ``pad_length - 1`` PADs after END, the id rows ``pad_length`` zeros)."""
        value = np.append(value, self.token.END)
        pad_length = self.max_input_length - len(value)
        if pad_length < 1:
            raise ValueError(f"{len(value) - 1} input tokens do not fit MAX_INPUT_LENGTH={self.max_input_length}")
        value = np.pad(value, (0, pad_length - 1), constant_values=self.token.PAD)
        out = {"input_value": value}
        for k, v in ids.items():
            out[k] = np.pad(v, (0, pad_length))
        out["input_mask"] = value == self.token.PAD
        return out

    def prepare_output_sequence(self, planks, attach):
        """reference line_data.py:85-109 / sideface_data.py:191-213."""
        value = quantize_values(planks, self.num_bits)
        value = np.append(value, self.token.END)
        value = np.pad(value, (0, self.max_output_length - len(value)), constant_values=self.token.PAD)
        label = np.pad(np.asarray(attach, dtype="long"), (0, self.max_output_length - len(attach)), constant_values=-1)
        ptr = label != -1
        label[ptr] += self.vocab_size
        label[~ptr] = value[~ptr]
        return {"output_value": value, "output_label": label, "output_mask": value == self.token.PAD}

    def _read_info(self, index):
        with open(os.path.join(self.root, self.info_files[index]), "r") as f:
            return json.loads(f.read())


class LineDataset(_TokenisedDataset):
    """reference plankassembly/datasets/line_data.py:12-142."""

    def prepare_input_sequence(self, lines, views, types):
        value, pos, coord, view, typ = self._sorted_tokens(lines, views, types)
        return self._pad_inputs(value, {"input_pos": pos, "input_coord": coord, "input_view": view, "input_type": typ})

    def __getitem__(self, index):
        info = self._read_info(index)
        lines = np.array(info["lines"], dtype="float")
        views = np.array(info["views"], dtype="long")
        types = np.array(info["types"], dtype="long")
        planks = np.array(info["coords"]).flatten()
        attach = np.array(info["attach"]).flatten()
        if self.augmentation and np.random.random() < self.aug_ratio:
            segs, views, types = add_noise([_segment_points(s) for s in info["svgs"]], list(views), list(types),
                                           self.noise_ratio, self.noise_length)
            lines = _bounds(segs)
        inputs = self.prepare_input_sequence(lines, views, types)
        outputs = self.prepare_output_sequence(planks, attach)
        return {"name": info["name"], **inputs, **outputs}


class SidefaceDataset(_TokenisedDataset):
    """reference plankassembly/datasets/sideface_data.py:85-254 minus the geometry (module docstring)."""

    def prepare_input_sequence(self, faces, views):
        if len(faces) != 0:
            value, pos, coord, view, _ = self._sorted_tokens(faces, views)
        else:                                                      # no side face detected: input = [END, PAD, ...]
            value = quantize_values(np.array(faces), self.num_bits)
            view = np.array(views, dtype="long")
            pos = np.zeros_like(view, dtype="long")
            coord = np.zeros_like(view, dtype="long")
        return self._pad_inputs(value, {"input_pos": pos, "input_coord": coord, "input_view": view})

    def __getitem__(self, index):
        info = self._read_info(index)
        if "faces" not in info or "faceviews" not in info:
            raise NotImplementedError(
                "side-face extraction from the line drawing (reference sideface_data.py:21-135, shapely polygonize + "
                "STRtree) is outside this package; store the extracted boxes in the info file as 'faces' "
                "[[xmin, ymin, xmax, ymax]] and 'faceviews' [0|1|2]")
        planks = np.array(info["coords"]).flatten()
        attach = np.array(info["attach"]).flatten()
        inputs = self.prepare_input_sequence(np.array(info["faces"], dtype="float").reshape(-1, 4), info["faceviews"])
        outputs = self.prepare_output_sequence(planks, attach)
        return {"name": info["name"], **inputs, **outputs}
