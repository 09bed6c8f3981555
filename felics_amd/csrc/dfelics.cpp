// dfelics -- decompresses a felics file to another image file (format from the output extension).
// Drop-in for the reference's src/bin/dfelics.rs.  Decoding is the host decoder of libfelics.
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/felics.h"
#include "cli_args.h"
#include "image_io.h"

static const char *variant_name(int rc) {  // `{:?}` of DecompressionError, dfelics.rs:39
    switch (rc) {
        case FELICS_E_IO: return "IoError(Custom { kind: UnexpectedEof, error: \"failed to fill whole buffer\" })";
        case FELICS_E_INVALID_VALUE: return "InvalidValue";
        case FELICS_E_VALUE_OVERFLOW: return "ValueOverflow";
        case FELICS_E_INVALID_DIMENSIONS: return "InvalidDimensions";
        case FELICS_E_INVALID_COLOR_TYPE: return "InvalidColorType";
        case FELICS_E_INVALID_PIXEL_DEPTH: return "InvalidPixelDepth";
        case FELICS_E_INVALID_SIGNATURE: return "InvalidSignature";
        default: return felics_strerror(rc);
    }
}

int main(int argc, char **argv) {
    CliArgs args = cli_parse(argc, argv, "dfelics", "Decompresses a felics file to another image file",
                             "The input felics file",
                             "The output file. The output format will be determined using the extension of the output file");
    FILE *f = fopen(args.input.c_str(), "rb");
    if (!f) {
        printf("Cannot open input file: %s\n", strerror(errno));  // dfelics.rs:29
        return 1;
    }
    std::vector<uint8_t> buf;
    uint8_t chunk[1 << 16];
    size_t got;
    while ((got = fread(chunk, 1, sizeof chunk, f)) > 0) buf.insert(buf.end(), chunk, chunk + got);
    fclose(f);

    felics_header hdr;
    int rc = felics_read_header(buf.data(), buf.size(), &hdr);
    imageio::Image img;
    if (rc == FELICS_OK) {
        img.width = hdr.width;
        img.height = hdr.height;
        img.channels = hdr.color_type == FELICS_COLOR_RGB ? 3 : 1;
        img.bits = hdr.pixel_depth == FELICS_DEPTH_16 ? 16 : 8;
        const uint64_t nbytes = (uint64_t)hdr.width * hdr.height * img.channels * (img.bits / 8);
        // a forged header must not make us allocate more than the stream could possibly describe
        if (nbytes > (uint64_t)buf.size() * 8 * 4096 + 64) {
            rc = FELICS_E_IO;
        } else {
            img.data.resize((size_t)nbytes);
            rc = felics_decompress(buf.data(), buf.size(), img.data.data(), img.data.size(), nullptr);
        }
    }
    if (rc != FELICS_OK) {
        printf("Error while decompressing the image: %s\n", variant_name(rc));  // dfelics.rs:39
        return 1;
    }
    std::string err = imageio::write_image(args.output, img);
    if (!err.empty()) {
        printf("Cannot save image: %s\n", err.c_str());  // dfelics.rs:55
        return 1;
    }
    return 0;
}
