/* vmm_edge.c -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * Device memory whose END is flush against unmapped address space: a virtual range of twice the rounded size is reserved
 * (hipMemAddressReserve), only its first half is backed and mapped (hipMemCreate / hipMemMap / hipMemSetAccess), and the
 * caller gets the pointer `end - bytes`.  A kernel that reads or writes one byte past `bytes` beyond the last page's
 * aligned line touches a page that is not present and the process dies with a GPU memory fault -- the device-side
 * equivalent of running the reference's fuzzers with their buffers against a guard page.  Used by
 * tests/test_gpu_edges.py to check that the batched calls never read behind src + srcSize / cSrc + cSrcSize
 * (SURVEY 8(b) ownership; programs/fuzzer.c:217-230 is the write-side check the reference itself makes).
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

typedef struct {
    void* va;            /* start of the reservation */
    size_t vaBytes;      /* size of the reservation (2 x mapped) */
    size_t mapped;       /* bytes backed and mapped at va */
    hipMemGenericAllocationHandle_t handle;
    void* user;          /* va + mapped - bytes */
} VmmEdge;

/* returns 0 on success; *out is filled */
__attribute__((visibility("default"))) int vmm_edge_alloc(size_t bytes, VmmEdge* out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
    if (e != hipSuccess) return (int)e;
    if (gran == 0) gran = 2u << 20;
    const size_t mapped = (bytes + gran - 1) / gran * gran + (bytes == 0 ? gran : 0);
    memset(out, 0, sizeof(*out));
    e = hipMemAddressReserve(&out->va, 2 * mapped, gran, NULL, 0);
    if (e != hipSuccess) return (int)e;
    out->vaBytes = 2 * mapped;
    e = hipMemCreate(&out->handle, mapped, &prop, 0);
    if (e != hipSuccess) { (void)hipMemAddressFree(out->va, out->vaBytes); return (int)e; }
    e = hipMemMap(out->va, mapped, 0, out->handle, 0);
    if (e != hipSuccess) { (void)hipMemRelease(out->handle); (void)hipMemAddressFree(out->va, out->vaBytes); return (int)e; }
    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof(acc));
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = dev;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(out->va, mapped, &acc, 1);
    if (e != hipSuccess) { (void)hipMemUnmap(out->va, mapped); (void)hipMemRelease(out->handle); (void)hipMemAddressFree(out->va, out->vaBytes); return (int)e; }
    out->mapped = mapped;
    out->user = (char*)out->va + mapped - bytes;
    return 0;
}

__attribute__((visibility("default"))) int vmm_edge_free(VmmEdge* v)
{
    if (!v->va) return 0;
    (void)hipDeviceSynchronize();
    (void)hipMemUnmap(v->va, v->mapped);
    (void)hipMemRelease(v->handle);
    (void)hipMemAddressFree(v->va, v->vaBytes);
    memset(v, 0, sizeof(*v));
    return 0;
}
