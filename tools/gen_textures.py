#!/usr/bin/env python3
"""Writes the small image files the texture fixtures use (scenes/textures/): procedural content, fixed seed, in the three
formats the host reads -- PNG (8-bit RGB, non-power-of-two; 8-bit grey; 16-bit RGBA; 4-bit palette), TGA (24-bit
uncompressed bottom-up, 8-bit mono RLE top-down) and PFM.  Checked in; regenerate with this script."""
import os, struct, zlib
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes", "textures")


def png(path, arr, ctype, depth, palette=None):
    h = arr.shape[0]
    w = arr.shape[1] // {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]   # arr rows hold interleaved channel samples
    if depth == 16:
        raw = b"".join(b"\x00" + arr[y].astype(">u2").tobytes() for y in range(h))
    elif depth == 8:
        raw = b"".join(b"\x00" + arr[y].astype(np.uint8).tobytes() for y in range(h))
    else:   # packed samples, one channel
        rows = []
        for y in range(h):
            bits = "".join(format(int(v), "0%db" % depth) for v in arr[y])
            bits += "0" * (-len(bits) % 8)
            rows.append(b"\x00" + bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8)))
        raw = b"".join(rows)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
    if palette is not None:
        data += chunk(b"PLTE", bytes(palette))
    data += chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b"")
    open(path, "wb").write(data)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(7)
    # colour pattern, 23 x 17 (Lanczos resampling to 32 x 32): soft blobs + stripes
    y, x = np.mgrid[0:17, 0:23]
    base = np.stack([0.5 + 0.5 * np.sin(x * 0.9 + y * 0.3), 0.5 + 0.5 * np.cos(y * 0.8 - x * 0.2), 0.5 + 0.5 * np.sin((x + y) * 0.45)], -1)
    base = np.clip(base * 0.8 + rng.random((17, 23, 3)) * 0.2, 0, 1)
    png(os.path.join(OUT, "color_23x17.png"), np.round(base * 255).reshape(17, 23 * 3), 2, 8)
    # grey height field 32 x 32 (bump / roughness maps), 8-bit grey PNG
    y, x = np.mgrid[0:32, 0:32]
    hgt = 0.5 + 0.25 * np.sin(x * 0.6) * np.cos(y * 0.4) + 0.25 * (((x // 4) + (y // 4)) % 2)
    png(os.path.join(OUT, "height_32.png"), np.round(np.clip(hgt, 0, 1) * 255), 0, 8)
    # alpha mask 16 x 16: a disc plus isolated texels, exact zeros outside (16-bit RGBA PNG: high byte is what survives)
    y, x = np.mgrid[0:16, 0:16]
    a = (((x - 7.5) ** 2 + (y - 7.5) ** 2) < 30).astype(np.float64)
    a[2, 3] = a[12, 13] = 1
    rgba = np.stack([a, a, a, np.ones_like(a)], -1)
    png(os.path.join(OUT, "mask_16.png"), np.round(rgba * 65535).reshape(16, 16 * 4), 6, 16)
    # 4-bit palette PNG 8 x 8
    pal = [int(v) for v in rng.integers(0, 256, 16 * 3)]
    png(os.path.join(OUT, "palette_8.png"), rng.integers(0, 16, (8, 8)), 3, 4, pal)
    # TGA 24-bit uncompressed, bottom-up (origin lower left), 16 x 8
    img = np.round(rng.random((8, 16, 3)) * 255).astype(np.uint8)
    hdr = struct.pack("<BBBHHBHHHHBB", 0, 0, 2, 0, 0, 0, 0, 0, 16, 8, 24, 0)
    open(os.path.join(OUT, "noise_16x8.tga"), "wb").write(hdr + img[::-1, :, ::-1].tobytes())
    # TGA 8-bit mono, RLE, top-down, 8 x 8: vertical ramp (every row one run)
    hdr = struct.pack("<BBBHHBHHHHBB", 0, 0, 11, 0, 0, 0, 0, 0, 8, 8, 8, 0x20)
    open(os.path.join(OUT, "ramp_8.tga"), "wb").write(hdr + b"".join(bytes([0x80 | 7, 20 + 30 * r]) for r in range(8)))
    # PFM 12 x 10 RGB with values above 1 (HDR reflectance scale), bottom-up rows as the format stores them
    pf = (rng.random((10, 12, 3)) * 1.5).astype("<f4")
    with open(os.path.join(OUT, "hdr_12x10.pfm"), "wb") as f:
        f.write(b"PF\n12 10\n-1.0\n")
        f.write(pf[::-1].tobytes())
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
