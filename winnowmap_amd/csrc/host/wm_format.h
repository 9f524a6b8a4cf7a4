// wm_format.h — PAF / SAM records (mm_write_paf3 / mm_write_sam3 / write_tags, src/format.c:280-548), single-segment reads.
#pragma once
#include "wm_core.h"
#include "wm_index.h"
#include "wm_mapper.h"
namespace wm {
void write_sam_header(std::string &s, const Index &idx, int argc, const char *const *argv);      // mm_write_sam_hdr, src/format.c:118
// appends every record of one read (one line each, '\n'-terminated), honouring --secondary=no, --sam-hit-only, --paf-no-hit
void write_read(std::string &s, const Index &idx, const ReadIn &rd, const ReadOut &out, int64_t flag);
}
