/*
 * TEST INFRASTRUCTURE ONLY — CPU restatement (double precision, scalar C) of the coordinate update the GPU path of the
 * reference runs after the forces: leap-frog, LINCS and SETTLE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline may call this; the product never does.
 *
 * Pinned by the reference's own known answers (tests/golden/update_refdata.json, extracted from
 * src/gromacs/mdlib/tests/refdata/WithParameters_{LeapFrogTest_SimpleIntegration,SettleTest_SatisfiesConstraints,
 * ConstraintsTest_SatisfiesConstraints}_*.xml with the inputs of leapfrogtestdata.cpp, settletestdata.cpp + watersystem.h
 * and constr.cpp) in tests/test_oracle_golden.py.
 */
#ifndef ORACLE_UPDATE_REF_H
#define ORACLE_UPDATE_REF_H

#ifdef __cplusplus
extern "C"
{
#endif

/* mdlib/update.cpp:343-396 (updateMDLeapfrogSimple) == mdlib/leapfrog_gpu_internal.cu:92-160.
 * x, v in/out [3 n]; xp out = x before the step; lambdas[numTempScaleValues] (0: none; 1: lambdas[0] for all; > 1: per group);
 * prDiag[3] = dtPressureCouple * diag(M) or NULL */
void oracle_leapfrog(int n, double* x, double* xp, double* v, const double* f, const double* invmass, double dt, int numTempScaleValues,
                     const double* lambdas, const unsigned short* groups, const double* prDiag);

/* mdlib/settle.cpp:111-151 (parameters) and :463-742 (settleTemplate) == mdlib/settle_gpu_internal.cu:92-372.
 * atoms[3 nsettle]; x before, xp after the unconstrained update (in/out); v in/out or NULL; virial[9] += or NULL;
 * box[9] row-major, pbcType 0 / 2 / 3 (pbcutil/pbc_aiuc.h:98-183) */
void oracle_settle(int nsettle, const int* atoms, double mO, double mH, double dOH, double dHH, const double* x, double* xp, double* v,
                   double invdt, double* virial, int pbcType, const double* box);

/* The LINCS algorithm as the GPU path of the reference runs it (mdlib/lincs_gpu_internal.cu:91-377; Hess et al. 1997:
 * mdlib/lincs.cpp:1023-1260 without the extra triangle recursions and without the angle warning).
 * iatoms[3 ncons] = (type, i, j); lengths[type]; x before, xp in/out, v in/out or NULL; virial[9] += or NULL */
void oracle_lincs(int ncons, const int* iatoms, const double* lengths, int natoms, const double* invmass, int numIterations, int expansionOrder,
                  const double* x, double* xp, double* v, double invdt, double* virial, int pbcType, const double* box);

#ifdef __cplusplus
}
#endif
#endif
