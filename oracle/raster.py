"""Oracle restatement of the box rasteriser (CPU, numpy + Python loops).

Follows reference ``src/utils/bb_to_img.py:5-20`` (``boxes_to_binary_map``): each box [2,4] is re-ordered to the corner
cycle 0,1,3,2, scaled ``*10 + 400``, drawn with ``PIL.ImageDraw.polygon(fill=1)`` into an 800x800 mode-'F' image and the
image is flipped vertically.

The polygon fill itself lives in a third-party dependency that is not part of /root/reference: Pillow (unpinned in the
reference's requirements.txt; 12.2.0 is what this image ships, and that is the version restated and pinned here).
``fill_polygon`` restates Pillow's ``ImagingDrawPolygon`` + ``polygon_generic`` (src/libImaging/Draw.c) for a
non-alpha 32-bit image: vertices truncated to int, float32 edge slopes, one scan line per integer y with the
intersections sorted and filled pairwise between round-half-up(left) and round-half-down(right), horizontal edges drawn
as lines, doubled intersections where an edge ends above the polygon's last row, and the "corner joining" adjustment of
an intersection that coincides with an earlier edge's end point.  All arithmetic is float32 with separate multiply and
add, exactly as the library performs it.

Pinned by tests/golden/box_raster.npz (outputs of the reference function itself) and, where Pillow is importable, checked
against Pillow directly on adversarial polygons.  Test infrastructure only -- see ``oracle/__init__.py``.
"""
import math

import numpy as np

F = np.float32
MAP = 800


def _round_up(f):        # Draw.c ROUND_UP: floor(f + 0.5) for f >= 0, in float32; the negative branch runs in double
    f = F(f)
    return int(math.floor(F(f + F(0.5)))) if f >= 0 else -int(math.floor(abs(float(f)) + 0.5))


def _round_down(f):      # Draw.c ROUND_DOWN: ceil(f - 0.5)
    f = F(f)
    return int(math.ceil(F(f - F(0.5)))) if f >= 0 else -int(math.ceil(abs(float(f)) - 0.5))


def _roundf(v):          # C roundf: half away from zero
    v = float(F(v))
    return F(math.floor(v + 0.5) if v >= 0 else -math.floor(-v + 0.5))


def _hline(img, xa, y, xb):
    """Draw.c hline32: clip to the image, no swap of the end points (an inverted span draws nothing)."""
    h, w = img.shape
    if y < 0 or y >= h or xa >= w or xb < 0:
        return
    xa, xb = max(xa, 0), min(xb, w - 1)
    if xa <= xb:
        img[y, xa:xb + 1] = 1


def fill_polygon(img, pts):
    """``ImageDraw.Draw(img).polygon(pts, fill=1)`` for integer vertices ``pts`` [(x, y), ...] on a 2-D array."""
    n, h = len(pts), img.shape[0]
    edges = []
    for i in range(n):
        (x0, y0), (x1, y1) = pts[i], pts[(i + 1) % n]
        if i == n - 1 and (x0, y0) == (x1, y1):
            continue                                   # closing edge only when the last vertex differs from the first
        dx = F(0.0) if y0 == y1 else F(F(x1 - x0) / F(y1 - y0))
        edges.append((min(x0, x1), max(x0, x1), min(y0, y1), max(y0, y1), x0, y0, dx))
    ymin, ymax, table = h - 1, 0, []
    for e in edges:
        ymin, ymax = min(ymin, e[2]), max(ymax, e[3])
        if e[2] == e[3]:
            _hline(img, e[0], e[2], e[1])              # horizontal edges are drawn, not scanned
        else:
            table.append(e)
    ymin, ymax = max(ymin, 0), min(ymax, h)

    def at(e, y):                                       # (y - y0) * dx + x0 in float32, multiply and add rounded apart
        return F(F(F(y - e[5]) * e[6]) + F(e[4]))

    for y in range(ymin, ymax + 1):
        xx = []
        for i, cur in enumerate(table):
            if not cur[2] <= y <= cur[3]:
                continue
            x = at(cur, y)
            if y == cur[3] and y < ymax:
                xx += [x, x]                            # an edge ending here counts twice ("consistent polygons")
                continue
            if (y == cur[3] or y == cur[2]) and cur[6] != 0:
                adj = y + 1 if y != cur[3] else y - 1
                for other in table[:i]:
                    if (y == other[2] or y == other[3]) and other[6] != 0 and _roundf(x) == _roundf(at(other, y)) \
                            and other[2] <= adj <= other[3]:
                        a, b = at(cur, adj), at(other, adj)
                        if x > F(a + F(1)) and x > F(b + F(1)):
                            x = F(_roundf(max(a, b)) + F(1))
                        elif F(a - F(1)) > x and F(b - F(1)) > x:
                            x = F(_roundf(min(a, b)) - F(1))
                        break
            xx.append(x)
        xx.sort()
        for i in range(1, len(xx), 2):
            _hline(img, _round_up(xx[i - 1]), y, _round_down(xx[i]))


def box_vertices(box):
    """[2,4] box (metres) -> four integer pixel vertices in drawing order.  bb_to_img.py:13-17 (+ Pillow's (int) cast)."""
    box = np.asarray(box)
    cyc = np.stack([box[:, 0], box[:, 1], box[:, 3], box[:, 2]]) * 10 + 400      # in the box's own dtype
    return [(int(float(p[0])), int(float(p[1]))) for p in cyc]                     # C (int): truncation toward zero


def boxes_to_binary_map(boxes):
    """[n,2,4] -> [800,800] float64 0/1 map, row 0 = top after the vertical flip.  bb_to_img.py:5-20."""
    img = np.zeros((MAP, MAP))
    for box in np.asarray(boxes):
        fill_polygon(img, box_vertices(box))
    return np.flip(img, 0)
