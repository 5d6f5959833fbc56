#include "compressor.hpp"

#include <fcntl.h>
#include <unistd.h>

#include <cstdlib>

#include "gpuar_hip.h"

namespace gip {

// The reference checks its packet geometry at construction time
// (src/compressor.cpp:8-16); here the same facts are compile-time.
static_assert(GPUAR_PACKET_BYTES % 16 == 0, "packets are read 16 bytes at a time");
static_assert(GPUAR_PACKET_BYTES < (1u << 14) - 257u, "model total must stay below 2^14 (coder precision 16 bits)");

Compressor::Compressor() {}

Compressor::~Compressor() { closeFiles(); }

void Compressor::openFiles(bool truncate_output) {
    openFile = std::fopen(openFileName.c_str(), "rb");
    if (!openFile) throw std::runtime_error("Can not open input file: " + openFileName);
    if (truncate_output) {
        saveFile = std::fopen(saveFileName.c_str(), "wb");
    } else {
        const int fd = ::open(saveFileName.c_str(), O_CREAT | O_WRONLY, 0666);
        saveFile = fd < 0 ? nullptr : ::fdopen(fd, "wb");
        if (fd >= 0 && !saveFile) ::close(fd);
    }
    if (!saveFile) {
        closeFiles();
        throw std::runtime_error("Can not open output file: " + saveFileName);
    }
}

size_t Compressor::getFileSize(FILE *stream) {
    const long at = std::ftell(stream);
    std::fseek(stream, 0, SEEK_END);
    const long end = std::ftell(stream);
    std::fseek(stream, at, SEEK_SET);
    return end < 0 ? 0 : static_cast<size_t>(end);
}

void Compressor::closeFiles() {
    if (saveFile) std::fclose(saveFile);
    if (openFile) std::fclose(openFile);
    saveFile = openFile = nullptr;
}

// src/compressor.cpp:28-44: `size` bytes of rand() output, 4 at a time
void Compressor::generateRandomFile(const size_t size) {
    saveFile = std::fopen(saveFileName.c_str(), "wb");
    if (!saveFile) throw std::runtime_error("Can not open output file: " + saveFileName);
    for (size_t i = 0; i < size; i += 4) {
        const int d = std::rand();
        if (std::fwrite(&d, sizeof d, 1, saveFile) != 1) {
            closeFiles();
            throw std::runtime_error("Write raw data to file failed");
        }
    }
    closeFiles();
}

}  // namespace gip
