"""felics_amd -- MI355X-native FELICS lossless image encoder (gfx950 HIP kernels behind a C ABI).

Layout: csrc/ (HIP kernels, C ABI, host decoder, command lines), api.py (host mirror of the
reference's public surface over ctypes), synth.py (the benchmark's synthetic frames).
"""
from .api import (ColorType, DecompressionError, Encoder, FelicsError, Header, PixelDepth,  # noqa: F401
                  compress, compress_image, decompress, decompress_image, decompress_with_header, read_header,
                  write_header)

__all__ = ["ColorType", "DecompressionError", "Encoder", "FelicsError", "Header", "PixelDepth", "compress",
           "compress_image", "decompress", "decompress_image", "decompress_with_header", "read_header", "write_header"]
