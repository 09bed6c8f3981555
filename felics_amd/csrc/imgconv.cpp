// imgconv -- reads an image file with the readers cfelics uses and writes it with the writers dfelics uses
// (format from the output extension).  CPU only: the front ends of the command lines on their own, for tests and
// for preparing corpora (bench/corpus.py).  Prints the layout the `image` crate would name (L8, Rgb16, ...).
#include <cstdio>

#include "cli_args.h"
#include "image_io.h"

int main(int argc, char **argv) {
    CliArgs args = cli_parse(argc, argv, "imgconv", "Converts an image file to another image format", "The input file",
                             "The output file. The output format will be determined using the extension of the output file");
    imageio::Image img;
    bool open_failed = false;
    std::string err = imageio::read_image(args.input, img, open_failed);
    if (!err.empty()) {
        printf("%s: %s\n", open_failed ? "Cannot open file" : "Cannot decode image", err.c_str());
        return 1;
    }
    printf("%s %ux%u\n", img.color_name().c_str(), img.width, img.height);
    err = imageio::write_image(args.output, img);
    if (!err.empty()) {
        printf("Cannot save image: %s\n", err.c_str());
        return 1;
    }
    return 0;
}
