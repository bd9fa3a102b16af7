/*
 * spk_assembly.h -- PETSc-free host assembler: the input generator of the hot
 * path.  Restates, on plain CSR arrays, what /root/reference/src/Discretization.c
 * does through DMDA/Mat/Vec: AssembleOperator_Laplace (:130-172),
 * AssembleRHS_Laplace (:174-227), ApplyBC_Laplace (:229-274) and the two
 * constraint assemblers the reference leaves as empty stubs (:277-290), whose
 * content is build-defined.  Host only (no GPU needed),
 * threaded over node lines.
 *
 * Grid: mx x my NODES (the reference's nx+1, ny+1; Discretization.c:17),
 * dof 2, natural ordering row = (j*mx + i)*2 + c, coordinates uniform on
 * [0,1]^2 (:25).  A slab is the row range [row_begin,row_end) of whole node
 * lines, as spk_partition_slab() deals them.
 */
#ifndef SPK_ASSEMBLY_H
#define SPK_ASSEMBLY_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Rows and stored non-zeros of the whole (0,0) block: 2*mx*my, 4(3mx-2)(3my-2)
 * (the 9-point x 2 x 2 pattern DMCreateMatrix preallocates; zeros stay stored). */
int SpkAssemblySizes(int mx, int my, int64_t *nrows, int64_t *nnz);
/* Stored non-zeros of rows [row_begin,row_end). */
int64_t SpkAssemblySlabNnz(int mx, int my, int64_t row_begin, int64_t row_end);

/* A (CSR, GLOBAL ascending column indices) and f for rows [row_begin,row_end).
 * apply_bc != 0: homogeneous Dirichlet on all four sides exactly as
 * MatZeroRowsColumns(A, bc, 1.0, NULL, NULL) + f_bc = 0 (Discretization.c:264,268).
 * rowptr has (row_end-row_begin)+1 entries and starts at 0.  f may be NULL.
 * nthreads <= 0: all hardware threads. */
int SpkAssembleOperator_Laplace(int mx, int my, int64_t row_begin, int64_t row_end, int32_t *rowptr,
                                int32_t *colidx, double *val, double *f, int apply_bc, int nthreads);

/* Build-defined constraint block B (4 rows) restricted to the columns
 * [row_begin,row_end), and g (4 values).  Rows: Ux mean, Uy mean, x-moment of
 * Ux, y-moment of Uy with lumped weights hx*hy on interior nodes; Dirichlet
 * columns dropped.  rowptr has 5 entries. */
int64_t SpkConstraintsSlabNnz(int mx, int my, int64_t row_begin, int64_t row_end);
int SpkAssembleOperator_Constraints(int mx, int my, int64_t row_begin, int64_t row_end, int32_t *rowptr,
                                    int32_t *colidx, double *val);
int SpkAssembleRHS_Constraints(double *g4);

/* ---- 3-D input generator: BUILD-DEFINED, NOT IN THE REFERENCE -------------------------------
 * The reference is 2-D only (#define DIM 2, include/Discretization.h:8).  BASELINE config 5 asks
 * for a 3-D grid and the reference's help string points to PETSc's ksp/ex42.c (main.c:1); these
 * routines carry the 2-D definitions above to Q1 hexahedra: mx x my x mz nodes on [0,1]^3, dof 3,
 * natural ordering row = ((k*my + j)*mx + i)*3 + c, 27-point x 3 x 3 pattern with zeros stored
 * (nnz = 9 (3mx-2)(3my-2)(3mz-2)), same truncated Gauss abscissa, stress form with
 * D = diag(2,2,2,1,1,1), body force (1,2,3), homogeneous Dirichlet on all six faces; six
 * constraint rows (component means + first moments).  A slab is a range of whole node PLANES
 * (3*mx*my rows each), as spk_partition_slab(mz, 3*mx*my, ...) deals them. */
int SpkAssemblySizes3D(int mx, int my, int mz, int64_t *nrows, int64_t *nnz);
int64_t SpkAssemblySlabNnz3D(int mx, int my, int mz, int64_t row_begin, int64_t row_end);
int SpkAssembleOperator_Laplace3D(int mx, int my, int mz, int64_t row_begin, int64_t row_end, int32_t *rowptr,
                                  int32_t *colidx, double *val, double *f, int apply_bc, int nthreads);
int64_t SpkConstraintsSlabNnz3D(int mx, int my, int mz, int64_t row_begin, int64_t row_end);
int SpkAssembleOperator_Constraints3D(int mx, int my, int mz, int64_t row_begin, int64_t row_end, int32_t *rowptr,
                                      int32_t *colidx, double *val);
int SpkAssembleRHS_Constraints3D(double *g6);
/* Optional divergence / pressure block for the 3-D grid (BUILD-DEFINED, in the manner of
 * PETSc's ksp/ex42.c that the reference's help string names, main.c:1): one constraint row per hexahedron e
 * (one constant pressure per element), B[e][(a,c)] = int_e dN_a/dx_c dV, Dirichlet columns dropped.  A general
 * sparse A10 block: m = (mx-1)(my-1)(mz-1) rows of <= 24 entries.  All m rows restricted to the node planes
 * [row_begin, row_end) the rank owns (column partition, like the other constraint rows); rowptr has m + 1 entries.
 * Note: with Dirichlet data on every face the rows sum to zero on the free columns (the constant-pressure mode). */
int64_t SpkDivergenceSlabNnz3D(int mx, int my, int mz, int64_t row_begin, int64_t row_end);
int SpkAssembleOperator_Divergence3D(int mx, int my, int mz, int64_t row_begin, int64_t row_end, int32_t *rowptr,
                                     int32_t *colidx, double *val);

/* Legacy-VTK ASCII output of the solution on the node grid: what WriteVTK(da_u, u,
 * "test.vtk") at /root/reference/src/SaddlePointProblem.c:22 is meant to produce.  The
 * reference's writer (Visulaization.c:3-67) emits points and polygons only and never the
 * field (its local vector is obtained at :27-28 and never filled); this one writes a
 * STRUCTURED_GRID with the node coordinates and POINT_DATA VECTORS U = (Ux, Uy, 0).
 * u has 2*mx*my entries in the natural ordering (j*mx + i)*2 + c. */
int SpkWriteVTK(int mx, int my, const double *u, const char *filename);

/* Element kernels (exposed for the known-answer tests): 8x8 stress-form
 * stiffness Ke[a*8+b] (Discretization.c:293-332) and load Fe (:334-374) for an
 * element with corner coordinates xe[8] = {x0,y0,x1,y1,x2,y2,x3,y3} in the
 * order (i,j) (i,j+1) (i+1,j+1) (i+1,j). */
int SpkFormStressOperatorQ12D(const double *xe, const double *coeff4, double *Ke64);
int SpkFormLaplaceRHSQ12D(const double *xe, double *Fe8);

#ifdef __cplusplus
}
#endif
#endif
