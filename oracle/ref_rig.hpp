// TEST INFRASTRUCTURE.  A minimal SPS / Slice / CodingStructure / CodingUnit around which a TransformUnit with a
// caller-supplied coefficient block can be handed to the reference's CABACWriter::residual_coding
// (cabac_writer.cpp:2424-2525).  Included by oracle/ref_harness.cpp and integration/reference_adapter_test.cpp after
// the reference headers (with private members opened, as those files do); holds no reference code.
#ifndef CABAC_REF_RIG_HPP
#define CABAC_REF_RIG_HPP
#include <cstdlib>
#include <memory>
#include <vector>

namespace {
template <class T> T *zeroed() { return reinterpret_cast<T *>(calloc(1, sizeof(T))); }  // never destroyed

struct ResidualRig {
  Common::SPS *sps = zeroed<Common::SPS>();
  Common::Slice *slice = zeroed<Common::Slice>();
  Common::CodingStructure *cs = zeroed<Common::CodingStructure>();
  Common::CodingUnit *cu = zeroed<Common::CodingUnit>();
  ResidualRig() {
    static bool rom = false;
    if (!rom) { Common::initROM(); rom = true; }
    sps->m_bitDepths.recon[0] = sps->m_bitDepths.recon[1] = 10;
    sps->m_log2MaxTbSize = 6;
    slice->m_pcSPS = sps;
    cs->sps = std::shared_ptr<const Common::SPS>(sps, [](const Common::SPS *) {});
    cs->slice = std::shared_ptr<Common::Slice>(slice, [](Common::Slice *) {});
    cu->cs = cs;
    cu->slice = slice;
  }
  // flags: bit0 dep_quant, bit1 sign_data_hiding, bit2 transform skip enabled in the SPS (max size 32),
  //        bit4 the block is transform-skip coded (mtsIdx = MTS_SKIP), bit5 BDPCM (cu.bdpcmMode / bdpcmModeChroma),
  //        bit6 SBT with MTS enabled in the SPS (sps.useMTS, cu.sbtInfo != 0): the zero-out of 32-wide / tall luma blocks,
  //        bits 8..15: if non-zero, extended_precision_processing with this bit depth.  comp: 0 Y, 1 Cb, 2 Cr.
  void make_tu(Common::TransformUnit &tu, std::vector<Common::TCoeff> &buf, int width, int height, int comp, int flags,
               const int32_t *coeff) {
    using namespace Common;
    slice->m_depQuantEnabledFlag = flags & 1;
    slice->m_signDataHidingEnabledFlag = (flags >> 1) & 1;
    sps->m_transformSkipEnabledFlag = (flags >> 2) & 1;
    sps->m_log2MaxTransformSkipBlockSize = 5;
    sps->m_MTS = (flags >> 6) & 1;
    cu->sbtInfo = (flags >> 6) & 1;
    const int depth = (flags >> 8) & 0xff;
    sps->m_spsRangeExtension.m_extendedPrecisionProcessingFlag = depth != 0;
    sps->m_bitDepths.recon[0] = sps->m_bitDepths.recon[1] = depth ? depth : 10;
    tu.initData();
    tu.chromaFormat = CHROMA_420;
    tu.blocks.clear();
    for (int c = 0; c < 3; c++)
      tu.blocks.push_back(CompArea(ComponentID(c), CHROMA_420, 0, 0, c == comp ? width : 0, c == comp ? height : 0));
    tu.cu = cu;
    tu.cs = cs;
    tu.chType = toChannelType(ComponentID(comp));
    buf.assign(coeff, coeff + (size_t)width * height);
    for (auto &p : tu.m_coeffs) p = nullptr;
    tu.m_coeffs[comp] = buf.data();
    tu.cbf[comp] = 1;
    tu.mtsIdx[comp] = (flags & 0x10) ? MTS_SKIP : MTS_DCT2_DCT2;
    cu->bdpcmMode = (flags & 0x20) ? 1 : 0;
    cu->bdpcmModeChroma = (flags & 0x20) ? 1 : 0;
  }
};
}  // namespace
#endif
