// kssd_dev.h -- the development build's instrumentation (libkssd_gpu_dev.so, `make -C public_kssd_amd tools`: -DKSSD_DEV) behind
// one set of macros, so that the product kernels read straight: in the shipping library every one of them expands to nothing.
//   KSSD_DEV_FIELD(decl)      a struct member that only the development build has (time-stamp buffers, A/B switches)
//   KSSD_DEV_STAMP(name)      const time stamp `name`, taken here (s_memrealtime: 10 ns ticks, one base for the whole chip)
//   KSSD_DEV_STAMP_CYC(name)  the same in shader cycles (readcyclecounter)
//   KSSD_DEV_VAR(name)        a time stamp to be taken later: KSSD_DEV_MARK(name) / KSSD_DEV_MARK_ONCE(name)
//   KSSD_DEV_DO(...)          statements of the development build only (writing the stamps out, an A/B branch)
#pragma once
#ifdef KSSD_DEV
#define KSSD_DEV_FIELD(decl) decl;
#define KSSD_DEV_STAMP(name) const unsigned long long name = __builtin_amdgcn_s_memrealtime()
#define KSSD_DEV_STAMP_CYC(name) const unsigned long long name = __builtin_readcyclecounter()
#define KSSD_DEV_VAR(name) unsigned long long name = 0
#define KSSD_DEV_MARK(name) name = __builtin_amdgcn_s_memrealtime()
#define KSSD_DEV_MARK_ONCE(name) do { if (!name) name = __builtin_amdgcn_s_memrealtime(); } while (0)
#define KSSD_DEV_DO(...) __VA_ARGS__
#else
#define KSSD_DEV_FIELD(decl)
#define KSSD_DEV_STAMP(name) do { } while (0)
#define KSSD_DEV_STAMP_CYC(name) do { } while (0)
#define KSSD_DEV_VAR(name) do { } while (0)
#define KSSD_DEV_MARK(name) do { } while (0)
#define KSSD_DEV_MARK_ONCE(name) do { } while (0)
#define KSSD_DEV_DO(...)
#endif
