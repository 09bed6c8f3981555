// image_io.h -- the small part of the `image` crate the reference's command lines rely on
// (src/bin/cfelics.rs:36-44, src/bin/dfelics.rs:45-52): read an image file into 8/16-bit gray or
// RGB samples, write one back in the format the file extension names.
// Formats: TIFF (II/MM, strips, chunky; uncompressed -- every file of the reference's image-suite and bench
// corpus -- or LZW / Deflate / PackBits, with or without the horizontal predictor), PNG (all colour types and
// bit depths, Adam7 included; read and write) and binary PNM (P5/P6).  zlib does the inflating / deflating.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace imageio {

struct Image {
    uint32_t width = 0, height = 0;
    int channels = 0;        // 1 gray, 2 gray+alpha, 3 rgb, 4 rgba
    int bits = 0;            // 8 or 16
    std::vector<uint8_t> data;  // row-major, interleaved, native-endian samples
    // name the `image` crate gives this layout (ColorType Debug): L8, L16, Rgb8, Rgb16, La8, Rgba8 ...
    std::string color_name() const;
};

// Returns "" on success, else a message. `open_failed` distinguishes "Cannot open file" from
// "Cannot decode image" (cfelics.rs:36-50).
std::string read_image(const std::string &path, Image &out, bool &open_failed);
std::string write_image(const std::string &path, const Image &img);

}  // namespace imageio
