/* mm_hostcopy.h -- device memory into a caller's host array through library-owned pinned bounce buffers (mm_hostcopy.hip) */
#ifndef MM_HOSTCOPY_H
#define MM_HOSTCOPY_H

#include <hip/hip_runtime.h>

#include <cstddef>

/* Copies `bytes` from device memory `d_src` (on `device`, ordered behind the work already queued on `stream`) into host
 * memory `dst` and returns when they are there.  Pageable destinations are filled through a ring of pinned chunks by
 * several host threads; pinned destinations and small copies take one hipMemcpyAsync.  The caller has `device` current. */
hipError_t mm_copy_to_host(void *dst, const void *d_src, size_t bytes, int device, hipStream_t stream);

#endif
