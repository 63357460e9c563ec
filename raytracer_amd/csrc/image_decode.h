// image_decode.h - colour-map files (PNG, BMP) to RGBA8; see image_decode.cpp.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace rtimg {

struct Image {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgba;      // row 0 = top, 4 bytes per pixel
    bool lfs_pointer = false;       // the file is a git-lfs pointer (decode_file returned false)
};

bool decode_file(const std::string& path, Image& out, std::string& err);

}  // namespace rtimg
