/* libtpsrhs_io.so -- TPS restart files (HDF5) to and from the operator's byNODES state vector.
 *
 * Replaces, for the solution state, M2ulPhyS::restart_files_hdf5 / read_restart_files_hdf5 / write_restart_files_hdf5
 * (src/io.cpp:43-193) with IOFamily::readPartitioned / writePartitioned (src/io.cpp:701-776): one file per rank,
 * attributes "iteration" (int), "time", "dt" (double), "order", "dimension" (int) and optionally "dofs_global"
 * on the root group, and the group "/solution" with one 1-D double dataset of NDofs entries per conserved variable:
 * "density", "rho-u", "rho-v", ["rho-w"], "rho-E", "rho-Y_<species>" for every active species and "rhoE_e" for a
 * two-temperature mixture (src/M2ulPhyS.cpp:1825-1852).  Dataset k holds entries [k*NDofs, (k+1)*NDofs) of the state
 * vector: the reference's Ordering::byNODES layout, which is the layout of tpsrhs_mult's x.  A library of its own: the
 * kernel library links no HDF5.  HOST buffers; nothing here touches the GPU.  [third party: HDF5 C library >= 1.10] */
#ifndef TPSRHS_IO_H_
#define TPSRHS_IO_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tpsrhs_restart_info {
  int iteration; /* attribute "iteration" */
  double time;   /* "time" */
  double dt;     /* "dt" */
  int order;     /* "order": polynomial order of the stored solution */
  int dimension; /* "dimension" */
  int64_t dofs_global; /* "dofs_global" of partitioned files, -1 when absent */
  int64_t ndofs;       /* entries of each dataset of this file */
} tpsrhs_restart_info;

/* Names of the /solution datasets in state-vector order (src/M2ulPhyS.cpp:1825-1852): nvel = 2 or 3 velocity
 * components, the names of the ACTIVE species in mixture order, two_temperature != 0 adds "rhoE_e".  Writes up to
 * `capacity` pointers into a static table of this library (valid until the next call from the same thread); returns
 * the number of names = num_equation, or -1. */
int tpsrhs_restart_variable_names(int nvel, int num_active_species, const char *const *species_names, int two_temperature,
                                  int capacity, const char **names);

/* Attributes and dataset size only (U may be sized from info->ndofs); 0 on success. */
int tpsrhs_restart_info_read(const char *path, tpsrhs_restart_info *info);

/* Reads dataset /solution/<names[k]> into U[k*ndofs ... ) for k < num_equation.  The file must hold datasets of
 * exactly `ndofs` entries (the reference asserts numInSoln == local_ndofs, src/io.cpp:763) and `order` must equal
 * info->order unless order < 0 (a file of another order: tpsrhs_restart_read_change_order).
 * 0 on success; tpsrhs_io_last_error() says what failed. */
int tpsrhs_restart_read(const char *path, int num_equation, int64_t ndofs, const char *const *names, int order, double *U,
                        tpsrhs_restart_info *info);

/* SERIALISED restart (`io/restartMode = singleFileRead*`, src/io.cpp:104-172, 460-530): ONE file holds the solution of
 * the unpartitioned mesh, every dataset with NDofs_global entries ordered by global element (a discontinuous space: the
 * dofs of global element g are entries [g * dofs_per_element, (g+1) * dofs_per_element) of each variable).  The reference
 * reads it on rank 0 and sends every rank its elements; here every rank opens the file read-only and takes the elements it
 * owns: `global_elements[e]` = id of local element e in the unpartitioned mesh (the partition the caller already has --
 * tpsrhs_mesh needs it for nothing else).  U: [num_equation][num_elements * dofs_per_element], byNODES.  A serialised file
 * is written with tpsrhs_restart_write from the global vector (dofs_global < 0: the reference omits that attribute there). */
int tpsrhs_restart_read_serial(const char *path, int num_equation, int64_t num_elements, int dofs_per_element,
                               const int64_t *global_elements, const char *const *names, int order, double *U,
                               tpsrhs_restart_info *info);

/* CHANGE OF ORDER on restart (`io/restartMode = variableP`; M2ulPhyS::read_restart_files_hdf5 with loadFromAuxSol,
 * src/io.cpp:174-193, and IOFamily::readChangeOrder, src/io.cpp:797-850): the file holds the solution of THIS rank's
 * `num_elements` elements at the polynomial order of its "order" attribute (info->order on return); every variable is read
 * into an auxiliary space of that order and interpolated to the operator's: per element u_new(x_i) = sum_a l_a^old(x_i) u_a
 * at the nodes x_i of the new order -- what mfem::GridFunction::ProjectGridFunction does between two nodal L2 spaces of one
 * basis type [third party: MFEM fem/gridfunc.cpp, FiniteElement::Project], applied here as a tensor product.
 * `basis_type`: tpsrhs_disc::basis_type (0 Gauss-Legendre, 1 Gauss-Lobatto nodes; the same in file and operator, as in the
 * reference, whose auxiliary collection is `fec->Clone(read_order)`).  U: [num_equation][num_elements * (order+1)^dim]. */
int tpsrhs_restart_read_change_order(const char *path, int num_equation, int64_t num_elements, int dim, int order, int basis_type,
                                     const char *const *names, double *U, tpsrhs_restart_info *info);

/* The reference's partitioned write of one rank (src/io.cpp:43-103, 701-724): truncates `path`. */
int tpsrhs_restart_write(const char *path, int num_equation, int64_t ndofs, const char *const *names, const double *U,
                         const tpsrhs_restart_info *info);

const char *tpsrhs_io_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
