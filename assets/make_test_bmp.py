#!/usr/bin/env python3
"""Writes assets/meshes/simple/test.bmp, the colour map of the "UV Test" scene.

Upstream keeps this file in git-lfs and only the pointer is available (256 x 256,
24-bit, 196662 bytes), so this is a stand-in of the same shape drawn here: an 8 x 8
grid whose cell colour encodes (u, v), with a white marker in the top-left corner.
"""
import os
import struct

W = H = 256
rows = []
for y in range(H):                      # y = 0 is the TOP row of the picture
    row = bytearray()
    for x in range(W):
        cx, cy = x // 32, y // 32
        r, g, b = 40 + cx * 30, 40 + cy * 30, 220 if (cx + cy) % 2 else 60
        if x % 32 == 0 or y % 32 == 0:
            r = g = b = 0
        if x < 16 and y < 16:
            r = g = b = 255
        row += bytes((b, g, r))
    rows.append(bytes(row))
data = b"".join(reversed(rows))         # BMP stores the bottom row first
header = b"BM" + struct.pack("<IHHI", 54 + len(data), 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, W, H, 1, 24, 0, len(data), 2835, 2835, 0, 0)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "meshes", "simple", "test.bmp")
with open(out, "wb") as f:
    f.write(header + data)
print(out, 54 + len(data), "bytes")
