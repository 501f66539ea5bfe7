// matrix_literal.h -- TEST HARNESS glue: the reference's object-at-a-time matrix arithmetic (Matrix.cpp:57-98,150-263) lives in the package's
// Matrix<T> (fhe-si_amd/host/fhesi_matrix.h), and Regression::Regress (Regression.h:102-149) in fhesi::Regression::Regress.  The drivers
// instantiate the same template with a plaintext ring type to get the plaintext regression the ciphertexts must decrypt to, and run Regress
// with the recording of Ciphertext operations on (batched device calls) and off (FHESI_EAGER / LazyCiphertexts() = false: every statement at
// once) as the checker of RegressBatched's explicit waves.
#pragma once
#include "../../fhe-si_amd/host/fhesi_matrix.h"

namespace fhesi {

template <class T> using LMatrix = Matrix<T>;
inline void RegressLiteral(const Regression& R, std::vector<Ciphertext>& theta, Ciphertext& det) { R.Regress(theta, det); }

}  // namespace fhesi
