// fiesta_amd/csrc/checkpoint.hpp -- raw dump / load of a map's device state to a file (host side).
//
// The reference has no checkpoint of its own (its state dies with the ROS node); SURVEY.md 5 lists dump/resume of the
// field as the auxiliary subsystem a long-running mapper needs.  The format is the device layout itself, written in
// pieces through a pinned staging buffer: a fixed header (magic, version, map mode, geometry), then named sections
// (byte count + payload).  A file only loads into a map created with the same mode, resolution, origin and grid.
#pragma once
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>

#include <string>

#include "../../include/fiesta_hip.h"
#include "common.hpp"

namespace fiesta {

class DevFile {
 public:
  // Writing goes to "<path>.tmp" and is renamed over <path> by finish(): a crash or a full disk never leaves a
  // half-written file under the real name.
  DevFile(const char *path, bool write, hipStream_t s) : s_(s), write_(write), path_(path), tmp_(std::string(path) + ".tmp") {
    f_ = fopen(write ? tmp_.c_str() : path, write ? "wb" : "rb");
    if (!f_) throw Error(FIESTA_HIP_ERR_INVALID, std::string("checkpoint: cannot open ") + path);
    if (!write) {
      struct stat st;
      if (fstat(fileno(f_), &st) == 0) size_ = (unsigned long long)st.st_size;
    }
    if (hipHostMalloc(&pin_, kChunk) != hipSuccess) {
      fclose(f_);
      throw Error(FIESTA_HIP_ERR_NOMEM, "checkpoint: no pinned staging buffer");
    }
  }
  ~DevFile() {
    if (pin_) (void)hipHostFree(pin_);
    if (f_) {
      fclose(f_);
      if (write_) (void)remove(tmp_.c_str());  // (finish() was never reached)
    }
  }
  DevFile(const DevFile &) = delete;
  DevFile &operator=(const DevFile &) = delete;

  void host(void *p, size_t bytes) {  // plain host data, written or read in place
    if (bytes == 0) return;
    const size_t k = write_ ? fwrite(p, 1, bytes, f_) : fread(p, 1, bytes, f_);
    if (k != bytes) throw Error(FIESTA_HIP_ERR_INVALID, write_ ? "checkpoint: short write" : "checkpoint: truncated file");
  }
  // a device array: its byte count is part of the file and must match on load
  void device(void *dev, size_t bytes) {
    unsigned long long n = bytes;
    host(&n, sizeof(n));
    if (n != bytes) throw Error(FIESTA_HIP_ERR_INVALID, "checkpoint: section size does not match this map");
    for (size_t off = 0; off < bytes; off += kChunk) {
      const size_t k = bytes - off < kChunk ? bytes - off : kChunk;
      if (write_) {
        FIESTA_HIP_CHECK(hipMemcpyAsync(pin_, (const char *)dev + off, k, hipMemcpyDeviceToHost, s_));
        FIESTA_HIP_CHECK(hipStreamSynchronize(s_));
        host(pin_, k);
      } else {
        host(pin_, k);
        FIESTA_HIP_CHECK(hipMemcpyAsync((char *)dev + off, pin_, k, hipMemcpyHostToDevice, s_));
        FIESTA_HIP_CHECK(hipStreamSynchronize(s_));
      }
    }
  }
  bool writing() const { return write_; }
  unsigned long long file_size() const { return size_; }  // (reading only)
  unsigned long long position() const { return (unsigned long long)ftell(f_); }
  void finish() {
    if (!write_) return;
    const bool ok = fflush(f_) == 0;
    const bool closed = fclose(f_) == 0;
    f_ = nullptr;
    if (!ok || !closed || rename(tmp_.c_str(), path_.c_str()) != 0) {
      (void)remove(tmp_.c_str());
      throw Error(FIESTA_HIP_ERR_INVALID, "checkpoint: flush / rename failed");
    }
  }

 private:
  static constexpr size_t kChunk = 64u << 20;
  FILE *f_ = nullptr;
  void *pin_ = nullptr;
  hipStream_t s_;
  bool write_;
  std::string path_, tmp_;
  unsigned long long size_ = 0;
};

struct CheckpointHeader {
  char magic[8];  // "FIESTAHP"
  uint32_t version, mode;
  int32_t grid[3], shard_lo[3], global[3];
  uint32_t pad;  // (explicit: the header is compared byte by byte)
  double res, org[3];
};
static_assert(sizeof(CheckpointHeader) == 8 + 8 + 36 + 4 + 32, "checkpoint header has no implicit padding");
inline void checkpoint_header(DevFile &f, uint32_t mode, const Geom &g) {
  CheckpointHeader h, mine;
  memset(&mine, 0, sizeof(mine));
  memcpy(mine.magic, "FIESTAHP", 8);
  mine.version = 5;  // 5: the late-observation marks and their counter (dense_map.hpp, C_LATE); 4: three list counters
  mine.mode = mode;
  mine.grid[0] = g.nx, mine.grid[1] = g.ny, mine.grid[2] = g.nz;
  if (mode == FIESTA_HIP_MODE_ARRAY) mine.shard_lo[0] = g.gx0, mine.shard_lo[1] = g.gy0, mine.shard_lo[2] = g.gz0;
  mine.global[0] = g.GX, mine.global[1] = g.GY, mine.global[2] = g.GZ;
  mine.res = g.res;
  for (int k = 0; k < 3; ++k) mine.org[k] = g.org[k];
  memcpy(&h, &mine, sizeof(h));
  f.host(&h, sizeof(h));
  if (!f.writing() && memcmp(&h, &mine, sizeof(h)) != 0) {  // say WHICH part differs
    if (memcmp(h.magic, mine.magic, 8) != 0) throw Error(FIESTA_HIP_ERR_INVALID, "checkpoint: not a fiesta_hip checkpoint (bad magic)");
    if (h.version != mine.version)
      throw Error(FIESTA_HIP_ERR_INVALID, "checkpoint: format version " + std::to_string(h.version) + ", this library reads version " +
                                              std::to_string(mine.version) + " only (older files cannot be migrated: re-save them with the library that wrote them)");
    if (h.mode != mine.mode) throw Error(FIESTA_HIP_ERR_INVALID, "checkpoint: written by a map of the other mode (array / hash)");
    throw Error(FIESTA_HIP_ERR_INVALID, "checkpoint: written by a map of another geometry (grid, shard box, resolution or origin)");
  }
}

}  // namespace fiesta
