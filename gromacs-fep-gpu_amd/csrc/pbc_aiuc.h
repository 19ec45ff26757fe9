/*
 * Minimum-image displacement with the box folded into nine floats ("all in unit cell": both positions lie in the unit cell
 * or next to it) — pbcutil/pbc_aiuc.h:67-183, device form pbcutil/pbc_aiuc_cuda.cuh:70-128.
 */
#ifndef NBNXM_PBC_AIUC_H
#define NBNXM_PBC_AIUC_H

#include <hip/hip_runtime.h>

namespace nbnxm_hip
{

struct PbcAiuc /* pbcutil/pbc_aiuc.h:67-96 */
{
    float invBoxDiagZ, boxZX, boxZY, boxZZ, invBoxDiagY, boxYX, boxYY, invBoxDiagX, boxXX;
};

/* setPbcAiuc (pbcutil/pbc_aiuc.h:98-140): pbcType 0 none, 2 xy, 3 xyz; dimensions without PBC get a zero inverse,
 * which makes their shift 0.  box: row-major 3x3. */
inline PbcAiuc makePbcAiuc(int pbcType, const float* box)
{
    const int npbcdim = (pbcType == 3) ? 3 : ((pbcType == 2) ? 2 : 0);
    PbcAiuc   pbc;
    pbc.invBoxDiagZ = (npbcdim > 2) ? 1.0F / box[8] : 0.0F;
    pbc.invBoxDiagY = (npbcdim > 1) ? 1.0F / box[4] : 0.0F;
    pbc.invBoxDiagX = (npbcdim > 0) ? 1.0F / box[0] : 0.0F;
    pbc.boxZX       = (npbcdim > 2) ? box[6] : 0.0F;
    pbc.boxZY       = (npbcdim > 2) ? box[7] : 0.0F;
    pbc.boxZZ       = (npbcdim > 2) ? box[8] : 0.0F;
    pbc.boxYX       = (npbcdim > 1) ? box[3] : 0.0F;
    pbc.boxYY       = (npbcdim > 1) ? box[4] : 0.0F;
    pbc.boxXX       = (npbcdim > 0) ? box[0] : 0.0F;
    return pbc;
}

__device__ __forceinline__ float3 pbcDxAiuc(const PbcAiuc& pbc, float3 a, float3 b)
{
    float3      dx  = make_float3(a.x - b.x, a.y - b.y, a.z - b.z);
    const float shz = rintf(dx.z * pbc.invBoxDiagZ);
    dx.x -= shz * pbc.boxZX;
    dx.y -= shz * pbc.boxZY;
    dx.z -= shz * pbc.boxZZ;
    const float shy = rintf(dx.y * pbc.invBoxDiagY);
    dx.x -= shy * pbc.boxYX;
    dx.y -= shy * pbc.boxYY;
    const float shx = rintf(dx.x * pbc.invBoxDiagX);
    dx.x -= shx * pbc.boxXX;
    return dx;
}

} // namespace nbnxm_hip
#endif
