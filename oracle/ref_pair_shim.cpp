// ref_pair_shim.cpp -- ORACLE infrastructure (not the product).
// Appended by oracle/build_ref.sh AFTER the reference's two pair-kernel
// definitions; gives them a C ABI so tests can call the reference arithmetic
// through ctypes.  Contains no reference code.
extern "C" {

// -> mobilityUFRPY (reference src/c_rigid_obj.cpp:31).  out6 = xx,xy,xz,yy,yz,zz
int ref_mobilityUFRPY(double rx, double ry, double rz, double *out6, int i, int j,
                      double inv_a)
{
  mobilityUFRPY(rx, ry, rz, out6[0], out6[1], out6[2], out6[3], out6[4], out6[5], i, j,
                inv_a);
  return 0;
}

// -> mobilityUFSingleWallCorrection (reference src/c_rigid_obj.cpp:85).
// M9 row-major, updated in place.  Returns 2 when the reference throws.
int ref_mobilityUFSingleWallCorrection(double rx, double ry, double rz, double *M9,
                                       int i, int j, double hj)
{
  try {
    mobilityUFSingleWallCorrection(rx, ry, rz, M9[0], M9[1], M9[2], M9[3], M9[4], M9[5],
                                   M9[6], M9[7], M9[8], i, j, hj);
  } catch (const std::runtime_error &) {
    return 2;
  }
  return 0;
}

}  // extern "C"
