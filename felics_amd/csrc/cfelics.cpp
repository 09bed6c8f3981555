// cfelics -- compresses an image file to a felics file on the GPU.
// Drop-in for the reference's src/bin/cfelics.rs: same flags, same stdout lines, exit status 1 on
// failure.  The image is decoded on the host, encoded by libfelics on an MI355X, written to disk.
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/felics.h"
#include "cli_args.h"
#include "image_io.h"

int main(int argc, char **argv) {
    // libfelics runs its stages on HIP streams of their own: ask the ROCm runtime for enough hardware queues before it
    // starts (an application's choice, not the library's; a value already in the environment wins)
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    CliArgs args = cli_parse(argc, argv, "cfelics", "Compresses an image file to a felics file", "The input file",
                             "The output felics file");
    imageio::Image img;
    bool open_failed = false;
    std::string err = imageio::read_image(args.input, img, open_failed);
    if (!err.empty()) {
        printf("%s: %s\n", open_failed ? "Cannot open file" : "Cannot decode image", err.c_str());  // cfelics.rs:39,47
        return 1;
    }
    if (img.channels != 1 && img.channels != 3) {
        printf("Unsupported image format: %s\n", img.color_name().c_str());  // cfelics.rs:70
        return 1;
    }
    printf("Compressing %d-bit %s image...\n", img.bits, img.channels == 1 ? "grayscale" : "rgb");  // cfelics.rs:54-66

    const int color = img.channels == 3 ? FELICS_COLOR_RGB : FELICS_COLOR_GRAY;
    const int depth = img.bits == 16 ? FELICS_DEPTH_16 : FELICS_DEPTH_8;
    felics_ctx *ctx = nullptr;
    int rc = felics_ctx_create(args.device, &ctx);
    if (rc != FELICS_OK) {
        printf("Cannot compress image: %s\n", felics_strerror(rc));
        return 1;
    }
    std::vector<uint8_t> out(img.data.size() + img.data.size() / 2 + 64);
    size_t n = 0;
    rc = felics_compress(ctx, img.data.data(), img.width, img.height, color, depth, out.data(), out.size(), &n);
    if (rc == FELICS_E_BUFFER_TOO_SMALL) {
        out.resize(n);
        rc = felics_compress(ctx, img.data.data(), img.width, img.height, color, depth, out.data(), out.size(), &n);
    }
    if (rc != FELICS_OK) {
        const char *detail = felics_last_error(ctx);
        printf("Cannot compress image: %s%s%s\n", felics_strerror(rc), *detail ? ": " : "", detail);  // cfelics.rs:76
        felics_ctx_destroy(ctx);
        return 1;
    }
    felics_ctx_destroy(ctx);
    FILE *f = fopen(args.output.c_str(), "wb");  // File::create, cfelics.rs:28
    if (!f) {
        printf("Cannot compress image: %s\n", strerror(errno));
        return 1;
    }
    const bool ok = fwrite(out.data(), 1, n, f) == n;
    if (fclose(f) != 0 || !ok) {
        printf("Cannot compress image: write failed\n");
        return 1;
    }
    return 0;
}
