// libtpsrhs_io.so -- include/tpsrhs_io.h: TPS restart files over the HDF5 C library (src/io.cpp:43-193, 701-776).
#include "../../include/tpsrhs_io.h"

#include <hdf5.h>

#include <cstring>

#include "basis.hpp"
#include <string>
#include <vector>

namespace {
thread_local std::string g_err;
thread_local std::vector<std::string> g_names;

int fail(const std::string &m) {
  g_err = m;
  return 1;
}

struct Handle {  // closes what it holds, in reverse order of opening
  hid_t id = -1;
  herr_t (*close)(hid_t) = nullptr;
  Handle(hid_t i, herr_t (*c)(hid_t)) : id(i), close(c) {}
  ~Handle() {
    if (id >= 0 && close) close(id);
  }
  Handle(const Handle &) = delete;
  Handle &operator=(const Handle &) = delete;
  operator hid_t() const { return id; }
};

struct QuietErrors {  // HDF5 prints its error stack by default; this library reports through its own string
  H5E_auto2_t fn = nullptr;
  void *data = nullptr;
  QuietErrors() {
    H5Eget_auto2(H5E_DEFAULT, &fn, &data);
    H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);
  }
  ~QuietErrors() { H5Eset_auto2(H5E_DEFAULT, fn, data); }
};

template <class T>
bool read_attr(hid_t file, const char *name, hid_t type, T *value) {  // h5_read_attribute, src/utils.hpp:80-90
  if (H5Aexists(file, name) <= 0) return false;
  Handle a(H5Aopen(file, name, H5P_DEFAULT), H5Aclose);
  return a >= 0 && H5Aread(a, type, value) >= 0;
}
template <class T>
bool write_attr(hid_t file, const char *name, hid_t type, const T &value) {  // h5_save_attribute, src/utils.hpp:66-77
  Handle sp(H5Screate(H5S_SCALAR), H5Sclose);
  Handle a(H5Acreate2(file, name, type, sp, H5P_DEFAULT, H5P_DEFAULT), H5Aclose);
  return a >= 0 && H5Awrite(a, type, &value) >= 0;
}

int read_info(hid_t file, const char *first_dataset, tpsrhs_restart_info *info) {
  long long dg = -1;
  if (!read_attr(file, "iteration", H5T_NATIVE_INT, &info->iteration)) return fail("restart file: attribute 'iteration' missing");
  if (!read_attr(file, "time", H5T_NATIVE_DOUBLE, &info->time)) return fail("restart file: attribute 'time' missing");
  if (!read_attr(file, "dt", H5T_NATIVE_DOUBLE, &info->dt)) return fail("restart file: attribute 'dt' missing");
  if (!read_attr(file, "order", H5T_NATIVE_INT, &info->order)) return fail("restart file: attribute 'order' missing");
  if (!read_attr(file, "dimension", H5T_NATIVE_INT, &info->dimension)) info->dimension = 0;  // older files
  int dgi = -1;
  info->dofs_global = read_attr(file, "dofs_global", H5T_NATIVE_INT, &dgi) ? dgi : dg;
  info->ndofs = -1;
  Handle d(H5Dopen2(file, first_dataset, H5P_DEFAULT), H5Dclose);
  if (d < 0) return fail(std::string("restart file: dataset ") + first_dataset + " missing");
  Handle sp(H5Dget_space(d), H5Sclose);
  if (H5Sget_simple_extent_ndims(sp) != 1) return fail(std::string("restart file: ") + first_dataset + " is not one-dimensional");
  hsize_t n = 0;
  H5Sget_simple_extent_dims(sp, &n, nullptr);  // get_variable_size_hdf5
  info->ndofs = static_cast<int64_t>(n);
  return 0;
}
}  // namespace

extern "C" {

const char *tpsrhs_io_last_error(void) { return g_err.c_str(); }

int tpsrhs_restart_variable_names(int nvel, int num_active_species, const char *const *species_names, int two_temperature,
                                  int capacity, const char **names) {
  if ((nvel != 2 && nvel != 3) || num_active_species < 0 || (num_active_species > 0 && !species_names) || !names) {
    g_err = "tpsrhs_restart_variable_names: invalid argument";
    return -1;
  }
  g_names.clear();
  g_names.push_back("density");
  g_names.push_back("rho-u");
  g_names.push_back("rho-v");
  if (nvel == 3) g_names.push_back("rho-w");
  g_names.push_back("rho-E");
  for (int sp = 0; sp < num_active_species; sp++) g_names.push_back(std::string("rho-Y_") + species_names[sp]);
  if (two_temperature) g_names.push_back("rhoE_e");
  const int n = static_cast<int>(g_names.size());
  if (n > capacity) {
    g_err = "tpsrhs_restart_variable_names: capacity too small";
    return -1;
  }
  for (int i = 0; i < n; i++) names[i] = g_names[i].c_str();
  return n;
}

int tpsrhs_restart_info_read(const char *path, tpsrhs_restart_info *info) {
  if (!path || !info) return fail("tpsrhs_restart_info_read: NULL argument");
  QuietErrors quiet;
  Handle f(H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT), H5Fclose);
  if (f < 0) return fail(std::string("cannot open restart file ") + path);
  return read_info(f, "/solution/density", info);
}

int tpsrhs_restart_read(const char *path, int num_equation, int64_t ndofs, const char *const *names, int order, double *U,
                        tpsrhs_restart_info *info) {
  if (!path || !names || !U || num_equation < 1 || ndofs < 0) return fail("tpsrhs_restart_read: invalid argument");
  QuietErrors quiet;
  Handle f(H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT), H5Fclose);
  if (f < 0) return fail(std::string("cannot open restart file ") + path);
  tpsrhs_restart_info local;
  tpsrhs_restart_info *in = info ? info : &local;
  const std::string first = std::string("/solution/") + names[0];
  if (read_info(f, first.c_str(), in) != 0) return 1;
  if (in->ndofs != ndofs)  // assert((int)numInSoln == local_ndofs_), src/io.cpp:763
    return fail("restart file holds " + std::to_string(in->ndofs) + " entries per variable, the operator has " + std::to_string(ndofs));
  if (order >= 0 && in->order != order)
    return fail("restart file of polynomial order " + std::to_string(in->order) + ", operator of order " + std::to_string(order) +
                " (tpsrhs_restart_read_change_order interpolates)");
  for (int k = 0; k < num_equation; k++) {  // read_variable_data_hdf5 into data + index * numInSoln
    const std::string p = std::string("/solution/") + names[k];
    Handle d(H5Dopen2(f, p.c_str(), H5P_DEFAULT), H5Dclose);
    if (d < 0) return fail("restart file: dataset " + p + " missing");
    Handle sp(H5Dget_space(d), H5Sclose);
    hsize_t n = 0;
    if (H5Sget_simple_extent_ndims(sp) != 1 || H5Sget_simple_extent_dims(sp, &n, nullptr) < 0 || static_cast<int64_t>(n) != ndofs)
      return fail("restart file: dataset " + p + " has the wrong size");
    if (H5Dread(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, U + static_cast<int64_t>(k) * ndofs) < 0)
      return fail("restart file: reading " + p + " failed");
  }
  return 0;
}

int tpsrhs_restart_read_serial(const char *path, int num_equation, int64_t num_elements, int dofs_per_element,
                               const int64_t *global_elements, const char *const *names, int order, double *U,
                               tpsrhs_restart_info *info) {
  if (!path || !names || !U || num_equation < 1 || num_elements < 0 || dofs_per_element < 1 || (num_elements > 0 && !global_elements))
    return fail("tpsrhs_restart_read_serial: invalid argument");
  QuietErrors quiet;
  Handle f(H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT), H5Fclose);
  if (f < 0) return fail(std::string("cannot open restart file ") + path);
  tpsrhs_restart_info local;
  tpsrhs_restart_info *in = info ? info : &local;
  const std::string first = std::string("/solution/") + names[0];
  if (read_info(f, first.c_str(), in) != 0) return 1;
  if (order >= 0 && in->order != order)
    return fail("restart file of polynomial order " + std::to_string(in->order) + ", operator of order " + std::to_string(order) +
                " (tpsrhs_restart_read_change_order interpolates)");
  const int64_t nglob = in->ndofs, ndofs = num_elements * dofs_per_element;
  if (nglob % dofs_per_element != 0) return fail("serial restart file: " + std::to_string(nglob) + " entries are not whole elements");
  for (int64_t e = 0; e < num_elements; e++)
    if (global_elements[e] < 0 || (global_elements[e] + 1) * dofs_per_element > nglob)
      return fail("serial restart file: global element " + std::to_string(global_elements[e]) + " outside the file's " +
                  std::to_string(nglob / dofs_per_element) + " elements");
  std::vector<double> all(static_cast<size_t>(nglob));
  for (int k = 0; k < num_equation; k++) {  // read_variable_data_hdf5 on the whole variable, then the rank's elements (:482-507)
    const std::string p = std::string("/solution/") + names[k];
    Handle d(H5Dopen2(f, p.c_str(), H5P_DEFAULT), H5Dclose);
    if (d < 0) return fail("restart file: dataset " + p + " missing");
    Handle sp(H5Dget_space(d), H5Sclose);
    hsize_t n = 0;
    if (H5Sget_simple_extent_ndims(sp) != 1 || H5Sget_simple_extent_dims(sp, &n, nullptr) < 0 || static_cast<int64_t>(n) != nglob)
      return fail("restart file: dataset " + p + " has the wrong size");
    if (H5Dread(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, all.data()) < 0) return fail("restart file: reading " + p + " failed");
    double *out = U + static_cast<int64_t>(k) * ndofs;
    for (int64_t e = 0; e < num_elements; e++)
      std::memcpy(out + e * dofs_per_element, all.data() + global_elements[e] * dofs_per_element, sizeof(double) * dofs_per_element);
  }
  return 0;
}

int tpsrhs_restart_read_change_order(const char *path, int num_equation, int64_t num_elements, int dim, int order, int basis_type,
                                     const char *const *names, double *U, tpsrhs_restart_info *info) {
  if (!path || !names || !U || num_equation < 1 || num_elements < 0 || (dim != 2 && dim != 3) || order < 1 || order >= tpsrhs::MAXN1 ||
      (basis_type != 0 && basis_type != 1))
    return fail("tpsrhs_restart_read_change_order: invalid argument");
  QuietErrors quiet;
  Handle f(H5Fopen(path, H5F_ACC_RDONLY, H5P_DEFAULT), H5Fclose);
  if (f < 0) return fail(std::string("cannot open restart file ") + path);
  tpsrhs_restart_info local;
  tpsrhs_restart_info *in = info ? info : &local;
  const std::string first = std::string("/solution/") + names[0];
  if (read_info(f, first.c_str(), in) != 0) return 1;
  const int po = in->order;  // read_order: the file's own attribute (src/io.cpp:120)
  if (po < 1 || po >= tpsrhs::MAXN1) return fail("restart file of polynomial order " + std::to_string(po) + ": orders 1.." + std::to_string(tpsrhs::MAXN1 - 1));
  const int no = po + 1, nn = order + 1;
  int64_t npe_o = 1, npe_n = 1;
  for (int d = 0; d < dim; d++) {
    npe_o *= no;
    npe_n *= nn;
  }
  if (in->ndofs != num_elements * npe_o)  // assert((int)numInSoln == aux_dof), src/io.cpp:813
    return fail("restart file holds " + std::to_string(in->ndofs) + " entries per variable: not " + std::to_string(num_elements) +
                " elements of order " + std::to_string(po));
  // P[i][a] = l_a^old(x_i^new): the 1-D factor of FiniteElement::Project between two nodal tensor elements of one basis type
  double xo[tpsrhs::MAXN1], xn[tpsrhs::MAXN1], w[tpsrhs::MAXN1], P[tpsrhs::MAXN1 * tpsrhs::MAXN1];
  (basis_type == 0 ? tpsrhs::gauss_legendre01 : tpsrhs::gauss_lobatto01)(no, xo, w);
  (basis_type == 0 ? tpsrhs::gauss_legendre01 : tpsrhs::gauss_lobatto01)(nn, xn, w);
  for (int i = 0; i < nn; i++)
    for (int a = 0; a < no; a++) P[i * no + a] = tpsrhs::lagrange(xo, no, a, xn[i]);
  const int64_t ndofs = num_elements * npe_n;
  std::vector<double> old(static_cast<size_t>(in->ndofs));
  std::vector<double> t1(static_cast<size_t>(nn) * no * no), t2(static_cast<size_t>(nn) * nn * no);
  for (int k = 0; k < num_equation; k++) {
    const std::string p = std::string("/solution/") + names[k];
    Handle d(H5Dopen2(f, p.c_str(), H5P_DEFAULT), H5Dclose);
    if (d < 0) return fail("restart file: dataset " + p + " missing");
    Handle sp(H5Dget_space(d), H5Sclose);
    hsize_t n = 0;
    if (H5Sget_simple_extent_ndims(sp) != 1 || H5Sget_simple_extent_dims(sp, &n, nullptr) < 0 || static_cast<int64_t>(n) != in->ndofs)
      return fail("restart file: dataset " + p + " has the wrong size");
    if (H5Dread(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, old.data()) < 0) return fail("restart file: reading " + p + " failed");
    double *out = U + static_cast<int64_t>(k) * ndofs;
    const int nz_o = dim == 3 ? no : 1, nz_n = dim == 3 ? nn : 1;
    for (int64_t e = 0; e < num_elements; e++) {  // nodes of an element are lexicographic, x fastest
      const double *src = old.data() + e * npe_o;
      double *dst = out + e * npe_n;
      for (int c = 0; c < nz_o; c++)  // x: [c][b][a] -> [c][b][i]
        for (int b = 0; b < no; b++)
          for (int i = 0; i < nn; i++) {
            double s = 0.0;
            for (int a = 0; a < no; a++) s += P[i * no + a] * src[(c * no + b) * no + a];
            t1[(c * no + b) * nn + i] = s;
          }
      for (int c = 0; c < nz_o; c++)  // y: [c][b][i] -> [c][j][i]
        for (int j = 0; j < nn; j++)
          for (int i = 0; i < nn; i++) {
            double s = 0.0;
            for (int b = 0; b < no; b++) s += P[j * no + b] * t1[(c * no + b) * nn + i];
            t2[(c * nn + j) * nn + i] = s;
          }
      if (dim == 2) {
        std::memcpy(dst, t2.data(), sizeof(double) * npe_n);
      } else {
        for (int l = 0; l < nz_n; l++)  // z: [c][j][i] -> [l][j][i]
          for (int ji = 0; ji < nn * nn; ji++) {
            double s = 0.0;
            for (int c = 0; c < no; c++) s += P[l * no + c] * t2[c * nn * nn + ji];
            dst[l * nn * nn + ji] = s;
          }
      }
    }
  }
  return 0;
}

int tpsrhs_restart_write(const char *path, int num_equation, int64_t ndofs, const char *const *names, const double *U,
                         const tpsrhs_restart_info *info) {
  if (!path || !names || !U || !info || num_equation < 1 || ndofs < 0) return fail("tpsrhs_restart_write: invalid argument");
  QuietErrors quiet;
  Handle f(H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), H5Fclose);
  if (f < 0) return fail(std::string("cannot create restart file ") + path);
  bool ok = write_attr(f, "iteration", H5T_NATIVE_INT, info->iteration) && write_attr(f, "time", H5T_NATIVE_DOUBLE, info->time) &&
            write_attr(f, "dt", H5T_NATIVE_DOUBLE, info->dt) && write_attr(f, "order", H5T_NATIVE_INT, info->order) &&
            write_attr(f, "dimension", H5T_NATIVE_INT, info->dimension);
  if (ok && info->dofs_global >= 0) ok = write_attr(f, "dofs_global", H5T_NATIVE_INT, static_cast<int>(info->dofs_global));
  if (!ok) return fail("restart file: writing the attributes failed");
  Handle g(H5Gcreate2(f, "/solution", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), H5Gclose);
  if (g < 0) return fail("restart file: cannot create /solution");
  const hsize_t dims[1] = {static_cast<hsize_t>(ndofs)};
  Handle sp(H5Screate_simple(1, dims, nullptr), H5Sclose);
  for (int k = 0; k < num_equation; k++) {  // write_variable_data_hdf5(group, name, dataspace, data + index * dims[0])
    Handle d(H5Dcreate2(g, names[k], H5T_NATIVE_DOUBLE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), H5Dclose);
    if (d < 0 || H5Dwrite(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, U + static_cast<int64_t>(k) * ndofs) < 0)
      return fail(std::string("restart file: writing /solution/") + names[k] + " failed");
  }
  return 0;
}

}  // extern "C"
