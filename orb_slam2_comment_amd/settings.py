"""The part of the example harness that feeds the front-end: the settings file (cv::FileStorage YAML subset read by the
Tracking constructor, src/Tracking.cc:51-125) and the KITTI sequence layout of Examples/Monocular/mono_kitti.cc:127-157.
No OpenCV: scalars are parsed from the text, images are decoded by a small 8-bit grayscale PNG / PGM reader."""
import os
import struct
import zlib

import numpy as np

MONOCULAR, STEREO, RGBD = 0, 1, 2      # System::eSensor, include/System.h:46-50


def load_settings(path):
    """`key: scalar` entries of an OpenCV YAML settings file (e.g. Examples/Monocular/KITTI00-02.yaml) as a dict of
    int / float / str.  The `%YAML:1.0` directive, comments and `!!opencv-matrix` blocks are skipped."""
    out = {}
    in_block = False
    with open(path) as f:
        for raw in f:
            line = raw.split("#", 1)[0].rstrip()
            if not line.strip() or line.startswith("%"):
                continue
            if line[0] in " \t":            # continuation of a nested mapping (matrix rows/cols/dt/data)
                continue
            if ":" not in line:
                continue
            key, val = line.split(":", 1)
            key, val = key.strip(), val.strip()
            in_block = val.startswith("!!") or val == ""
            if in_block:
                continue
            try:
                out[key] = int(val)
            except ValueError:
                try:
                    out[key] = float(val)
                except ValueError:
                    out[key] = val.strip('"')
    return out


def extractor_args(settings):
    """(nFeatures, scaleFactor, nLevels, iniThFAST, minThFAST), src/Tracking.cc:112-116."""
    return (int(settings["ORBextractor.nFeatures"]), float(settings["ORBextractor.scaleFactor"]),
            int(settings["ORBextractor.nLevels"]), int(settings["ORBextractor.iniThFAST"]),
            int(settings["ORBextractor.minThFAST"]))


def make_extractors(settings, sensor, device=0):
    """The extractor objects the Tracking constructor creates (src/Tracking.cc:119-125): left always, right for
    stereo, and the 2*nFeatures initialisation extractor for monocular."""
    from .extractor import ORBextractor
    nf, sf, nl, ini, mn = extractor_args(settings)
    ex = {"left": ORBextractor(nf, sf, nl, ini, mn, device=device)}
    if sensor == STEREO:
        ex["right"] = ORBextractor(nf, sf, nl, ini, mn, device=device)
    if sensor == MONOCULAR:
        ex["ini"] = ORBextractor(2 * nf, sf, nl, ini, mn, device=device)
    return ex


def load_kitti_sequence(path_to_sequence, camera="image_0"):
    """LoadImages of Examples/Monocular/mono_kitti.cc:127-157: timestamps from times.txt (empty lines skipped), file
    names <sequence>/image_0/%06d.png.  Returns (filenames, timestamps)."""
    stamps = []
    with open(os.path.join(path_to_sequence, "times.txt")) as f:
        for s in f:
            if s.strip():
                stamps.append(float(s.split()[0]))
    names = [os.path.join(path_to_sequence, camera, "%06d.png" % i) for i in range(len(stamps))]
    return names, stamps


def read_gray_image(path):
    """8-bit grayscale image as a 2-D uint8 array.  Supports non-interlaced 8-bit grayscale PNG (what KITTI odometry
    ships), binary PGM (P5, maxval 255) and .npy."""
    if path.endswith(".npy"):
        a = np.load(path)
        if a.dtype != np.uint8 or a.ndim != 2:
            raise ValueError("%s: expected a 2-D uint8 array" % path)
        return np.ascontiguousarray(a)
    if path.endswith(".png"):
        try:                                        # C decoder when Pillow happens to be installed
            from PIL import Image
            with Image.open(path) as im:
                if im.mode == "L":
                    return np.ascontiguousarray(np.asarray(im, dtype=np.uint8))
        except Exception:                           # Pillow absent or unable to read it: the reader below decides
            pass
    return _read_gray_image_py(path)


def _read_gray_image_py(path):
    data = open(path, "rb").read()
    if data[:2] == b"P5":
        parts, pos = [], 2
        while len(parts) < 3:                       # width, height, maxval separated by whitespace / comments
            while data[pos:pos + 1].isspace():
                pos += 1
            if data[pos:pos + 1] == b"#":
                pos = data.index(b"\n", pos) + 1
                continue
            end = pos
            while not data[end:end + 1].isspace():
                end += 1
            parts.append(int(data[pos:end]))
            pos = end
        w, h, maxval = parts
        if maxval != 255:
            raise ValueError("%s: only 8-bit PGM is supported" % path)
        return np.frombuffer(data, np.uint8, w * h, pos + 1).reshape(h, w).copy()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("%s: not a PNG / PGM / npy file" % path)
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
        pos += 12 + n
    if hdr is None:
        raise ValueError("%s: no IHDR" % path)
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or ctype != 0 or interlace != 0:
        raise ValueError("%s: only non-interlaced 8-bit grayscale PNG is supported (depth %d, colour type %d)"
                         % (path, depth, ctype))
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, w + 1)
    out = np.zeros((h, w), np.uint8)
    prev = np.zeros(w, np.uint8)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:]
        if ft == 0:
            cur = line.copy()
        elif ft == 2:
            cur = line + prev                        # uint8 wrap-around is the PNG arithmetic
        elif ft == 1:
            cur = np.cumsum(line, dtype=np.uint64).astype(np.uint8)
        else:                                        # Average / Paeth depend on the pixel to the left: serial in x
            cur = np.zeros(w, np.uint8)
            left = 0
            up = prev.astype(np.int32)
            for x in range(w):
                if ft == 3:
                    pred = (left + int(up[x])) >> 1
                else:
                    ul = int(up[x - 1]) if x else 0
                    p = left + int(up[x]) - ul
                    pa, pb, pc = abs(p - left), abs(p - int(up[x])), abs(p - ul)
                    pred = left if (pa <= pb and pa <= pc) else (int(up[x]) if pb <= pc else ul)
                left = (int(line[x]) + pred) & 255
                cur[x] = left
        out[y] = cur
        prev = cur
    return out
