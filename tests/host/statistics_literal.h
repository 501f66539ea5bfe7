// statistics_literal.h -- TEST HARNESS (not part of the package; SURVEY.md section 2 marks Statistics.h out of scope): a caller of the
// class surface in the shape of the reference's Statistics (Statistics.h:12-177), used only to drive the recording Ciphertext of
// fhesi_engine.h with one more object-at-a-time control flow (moments, covariance) and compare recorded with at-once evaluation.  In an
// integration the reference's own Statistics.h compiles unmodified on the mirrored classes.  As in Regression (fhe-si_amd/host/fhesi_matrix.h), plaintexts are coefficient vectors, the slot COUNT sizes the list of
// automorphism keys (Statistics.h:17-26), and GenerateNoise (Statistics.h:163-174: EmbedInSlots, slot packing) is not applied -- the results
// are the unmasked moments.
#pragma once
#include "matrix_literal.h"

namespace fhesi {

class Statistics {
  const FHEcontext& context;
  FHESISecKey secretKey;
  FHESIPubKey publicKey;
  KeySwitchSI keySwitch;
  std::vector<KeySwitchSI> autoKeySwitch;
  std::vector<unsigned> autoK;
  LMatrix<Ciphertext> data;
  std::vector<Ciphertext> nElems;

  void SumBatchedData(Ciphertext& batchedData) const {           // Statistics.h:148-161
    for (size_t i = 0; i < autoKeySwitch.size(); ++i) {
      Ciphertext tmp = batchedData;
      tmp >>= (long)autoK[i];
      autoKeySwitch[i].ApplyKeySwitch(tmp);
      batchedData += tmp;
    }
  }

 public:
  Statistics(const FHEcontext& c) : context(c), secretKey(c), publicKey(secretKey), keySwitch(secretKey), data(Ciphertext(c)) {   // Statistics.h:14-27
    unsigned k = c.Generator();
    unsigned nSlots = UsableSlots(c.zMstar.M(), (unsigned long)c.ModulusP().to_long(), c.zMstar.phiM());
    while (nSlots > 1) {
      autoKeySwitch.push_back(KeySwitchSI(secretKey, k));
      autoK.push_back(k);
      nSlots >>= 1;
      k = (unsigned)(((unsigned long)k * k) % c.zMstar.M());
    }
  }
  const std::vector<unsigned>& AutomorphismExponents() const { return autoK; }

  void AddData(const Matrix<Plaintext>& blocks, const std::vector<Plaintext>& blockSizes) {   // Statistics.h:29-41; the blocks of a call are encrypted in ONE device call
    std::vector<Plaintext> flat;
    for (unsigned i = 0; i < blocks.NumRows(); ++i) { for (auto& pt : blocks[i]) flat.push_back(pt); flat.push_back(blockSizes[i]); }
    std::vector<Ciphertext> enc;
    publicKey.EncryptBatch(enc, flat);
    size_t at = 0;
    for (unsigned i = 0; i < blocks.NumRows(); ++i) {
      std::vector<Ciphertext> encExample(enc.begin() + at, enc.begin() + at + blocks[i].size());
      at += blocks[i].size();
      data.AddRow(encExample);
      nElems.push_back(enc[at++]);
    }
  }
  void Clear() { data.Clear(); nElems.clear(); }

  void ComputeNthMoment(std::vector<Ciphertext>& moment, Ciphertext& denom, unsigned n) {   // Statistics.h:48-85
    if (n < 1 || n > 2) return;                                    // not supported by the reference either
    moment.assign(data.NumCols(), Ciphertext(context));
    denom = nElems[0];
    for (unsigned j = 0; j < data.NumCols(); ++j) {
      moment[j] = data(0, j);
      if (n == 2) moment[j] *= moment[j];
      for (unsigned i = 1; i < data.NumRows(); ++i) {
        if (j == 0) denom += nElems[i];
        Ciphertext tmp = data(i, j);
        if (n == 2) tmp *= tmp;
        moment[j] += tmp;
      }
      if (n == 2) keySwitch.ApplyKeySwitch(moment[j]);
      SumBatchedData(moment[j]);
    }
  }

  void ComputeCovariance(LMatrix<Ciphertext>& cov, std::vector<Ciphertext>& mu, Ciphertext& n, Ciphertext& n2) {   // Statistics.h:87-133
    ComputeNthMoment(mu, n, 1);
    Ciphertext dummy(context);
    LMatrix<Ciphertext> muMat(dummy);
    muMat.AddRow(mu);
    muMat.Transpose();
    muMat.MultByTranspose();
    for (unsigned i = 0; i < muMat.NumRows(); ++i)                 // symmetric: the upper triangle
      for (unsigned j = i; j < muMat.NumCols(); ++j) { keySwitch.ApplyKeySwitch(muMat(i, j)); muMat(i, j) *= -1; }
    cov = data;
    cov.Transpose();
    cov.MultByTranspose();
    for (unsigned i = 0; i < cov.NumRows(); ++i)
      for (unsigned j = i; j < cov.NumCols(); ++j) {
        keySwitch.ApplyKeySwitch(cov(i, j));
        SumBatchedData(cov(i, j));
        cov(i, j) *= n;
        keySwitch.ApplyKeySwitch(cov(i, j));
        cov(i, j) += muMat(i, j);
        cov(j, i) = cov(i, j);
      }
    n2 = n;
    n2 *= n2;
    keySwitch.ApplyKeySwitch(n2);
  }

  FHESISecKey& GetSecretKey() { return secretKey; }
  FHESIPubKey& GetPublicKey() { return publicKey; }
};

}  // namespace fhesi
