// matrix_literal.h -- TEST HARNESS: the reference's object-at-a-time matrix arithmetic (Matrix.cpp:57-98,150-263) and
// Regression::Regress control flow (Regression.h:102-149), restated on the mirrored classes.  It is the checker the device waves of
// fhesi::Regression::RegressBatched are compared with (bit-identical ciphertexts) and, instantiated with a plaintext ring type, the
// plaintext regression they must decrypt to.  With T = Ciphertext every statement below is RECORDED by the mirror's Ciphertext and
// evaluated in batched device calls (fhesi_engine.h) -- which is what an integration gets from the reference's own Matrix.cpp /
// Regression.h compiled unmodified on the mirrored classes.  Not part of the package: nothing under fhe-si_amd/ includes it.
#pragma once
#include "../../fhe-si_amd/host/fhesi_matrix.h"

namespace fhesi {

template <class T>
class LMatrix : public Matrix<T> {
  using Matrix<T>::dummy; using Matrix<T>::mat; using Matrix<T>::transpose; using Matrix<T>::ElemAt;
 public:
  using Matrix<T>::NumRows; using Matrix<T>::NumCols;
  LMatrix() : Matrix<T>() {}
  LMatrix(const T& d) : Matrix<T>(d) {}
  LMatrix(unsigned nRows, unsigned nCols, const T& d) : Matrix<T>(nRows, nCols, d) {}
  LMatrix(unsigned nRows, unsigned nCols) : Matrix<T>(nRows, nCols) {}
  LMatrix(const Matrix<T>& m) : Matrix<T>(m) {}
 private:
  // Laplace expansion along the first unused row (Matrix.cpp:227-263); `reduce` runs on every partial determinant of size >= 2
  void Determinant(T& det, std::vector<bool>& usedRows, std::vector<bool>& usedCols, unsigned dim, std::function<void(T&)> reduce) const {
    const unsigned matDim = NumRows();
    unsigned row = 0;
    while (usedRows[row]) ++row;
    bool negative = false, first = true;
    for (unsigned col = 0; col < matDim; ++col) {
      if (usedCols[col]) continue;
      if (dim == 1) { det = ElemAt(row, col); return; }
      T term = ElemAt(row, col);
      if (negative) term *= -1;
      negative = !negative;
      usedRows[row] = usedCols[col] = true;
      T minor(dummy);
      Determinant(minor, usedRows, usedCols, dim - 1, reduce);
      usedRows[row] = usedCols[col] = false;
      term *= minor;
      if (first) { det = term; first = false; } else det += term;
    }
    if (reduce) reduce(det);
  }

 public:

  LMatrix& operator+=(const LMatrix& o) { for (unsigned i = 0; i < NumRows(); ++i) for (unsigned j = 0; j < NumCols(); ++j) ElemAt(i, j) += o(i, j); return *this; }
  LMatrix operator+(const LMatrix& o) const { LMatrix r = *this; r += o; return r; }
  LMatrix& operator-=(const LMatrix& o) {                                                    // Matrix.cpp:30-40
    for (unsigned i = 0; i < NumRows(); ++i) for (unsigned j = 0; j < NumCols(); ++j) { T t = o(i, j); t *= -1; ElemAt(i, j) += t; }
    return *this;
  }
  LMatrix operator-(const LMatrix& o) const { LMatrix r = *this; r -= o; return r; }

  LMatrix& operator*=(LMatrix& o) {                                                          // Matrix.cpp:57-79
    if (mat.empty()) return *this;
    LMatrix prod(NumRows(), o.NumCols(), dummy);
    for (unsigned i = 0; i < NumRows(); ++i)
      for (unsigned j = 0; j < o.NumCols(); ++j) {
        prod(i, j) = ElemAt(i, 0);
        prod(i, j) *= o(0, j);
        for (unsigned k = 1; k < NumCols(); ++k) { T t = ElemAt(i, k); t *= o(k, j); prod(i, j) += t; }
      }
    std::swap(prod.mat, mat);
    transpose = false;
    return *this;
  }
  LMatrix& operator*=(std::vector<T>& v) {                                                  // Matrix.cpp:81-98 (the entries are multiplied in place first)
    if (mat.empty()) return *this;
    LMatrix prod(NumRows(), 1, dummy);
    for (unsigned i = 0; i < NumRows(); ++i) {
      ElemAt(i, 0) *= v[0];
      prod(i, 0) = ElemAt(i, 0);
      for (unsigned j = 1; j < NumCols(); ++j) { ElemAt(i, j) *= v[j]; prod(i, 0) += ElemAt(i, j); }
    }
    std::swap(mat, prod.mat);
    transpose = false;
    return *this;
  }
  LMatrix& operator*=(T& s) { for (unsigned i = 0; i < NumRows(); ++i) for (unsigned j = 0; j < NumCols(); ++j) ElemAt(i, j) *= s; return *this; }
  LMatrix operator*(LMatrix& o) const { LMatrix r = *this; r *= o; return r; }
  LMatrix operator*(std::vector<T>& v) const { LMatrix r = *this; r *= v; return r; }

  void MultByTranspose() {                                                                 // Matrix.cpp:150-174: upper triangle, mirrored
    if (mat.empty()) return;
    LMatrix prod(NumRows(), NumRows(), dummy);
    for (unsigned i = 0; i < NumRows(); ++i)
      for (unsigned j = i; j < NumRows(); ++j) {
        prod(i, j) = ElemAt(i, 0);
        prod(i, j) *= ElemAt(j, 0);
        for (unsigned k = 1; k < NumCols(); ++k) { T t = ElemAt(i, k); t *= ElemAt(j, k); prod(i, j) += t; }
        if (i != j) prod(j, i) = prod(i, j);
      }
    std::swap(prod.mat, mat);
    transpose = false;
  }
  void Determinant(T& det, std::function<void(T&)> reduce = nullptr) const {
    std::vector<bool> usedRows(NumRows()), usedCols(NumRows());
    Determinant(det, usedRows, usedCols, NumRows(), reduce);
  }
  void Invert(T& det, std::function<void(T&)> reduce = nullptr) {                          // Matrix.cpp:182-216: adjugate, then det from its first column
    const unsigned dim = NumRows();
    LMatrix adj(dim, dim, dummy);
    std::vector<bool> usedRows(dim), usedCols(dim);
    for (unsigned i = 0; i < dim; ++i)
      for (unsigned j = 0; j < dim; ++j) {
        usedRows[i] = usedCols[j] = true;
        Determinant(adj(j, i), usedRows, usedCols, dim - 1, reduce);
        usedRows[i] = usedCols[j] = false;
        if ((i + j) % 2 == 1) adj(j, i) *= -1;
      }
    det = ElemAt(0, 0);
    det *= adj(0, 0);
    for (unsigned i = 1; i < dim; ++i) { T t = ElemAt(0, i); t *= adj(i, 0); det += t; }
    if (reduce) reduce(det);
    std::swap(adj.mat, mat);
    transpose = false;
  }
};

// Regression::Regress, the reference's control flow one Ciphertext object at a time (Regression.h:102-149 without the GenerateNoise
// masking, which needs slot packing -- see fhesi_matrix.h)
inline void RegressLiteral(const Regression& R, std::vector<Ciphertext>& theta, Ciphertext& det) {
  LMatrix<Ciphertext> dataCopy(R.Data());
  std::vector<Ciphertext> lab = R.labels;
  dataCopy.Transpose();
  LMatrix<Ciphertext> last = dataCopy * lab;
  dataCopy.MultByTranspose();
  auto processFunc = [&R](Ciphertext& ct) { R.KeySwitch().ApplyKeySwitch(ct); R.SumBatchedDataObject(ct); };
  last.MapAll(processFunc);
  dataCopy.MapAll(processFunc);
  if (R.Data().NumCols() == 1) { det = dataCopy(0, 0); theta.assign(1, last(0, 0)); return; }
  dataCopy.Invert(det, [&R](Ciphertext& ct) { R.KeySwitch().ApplyKeySwitch(ct); });
  dataCopy *= last;
  dataCopy.MapAll([&R](Ciphertext& ct) { R.KeySwitch().ApplyKeySwitch(ct); });
  theta.assign(dataCopy.NumRows(), Ciphertext(R.Context()));
  for (unsigned i = 0; i < dataCopy.NumRows(); ++i) theta[i] = dataCopy(i, 0);
}

}  // namespace fhesi
