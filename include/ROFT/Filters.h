// Filters.h -- the C++ facade over the C ABI (include/roft_engine.h): one header per reference header, same class names,
// constructor signatures and virtuals (over the stand-in third-party types of Compat.h), every arithmetic member
// forwarding to the ABI.
//
//   reference class (src/roft-lib/include/ROFT/...)          header here                          C ABI behind it
//   SpatialVelocityModel + bfl::KFPrediction                 SpatialVelocityModel.h               roft_kf_predict
//   ImageOpticalFlowMeasurement<T> (.hpp:43-128)             ImageOpticalFlowMeasurement.hpp      roft_flow_measurement
//   SKFCorrection (SKFCorrection.h:23-52)                    SKFCorrection.h                      roft_skf_correct
//   ImageSegmentationOFAidedSource<T> (.hpp:34-92)           ImageSegmentationOFAidedSource.hpp   roft_mask_propagate
//   CartesianQuaternionModel + bfl::UKFPrediction            CartesianQuaternionModel.h           roft_ukf_predict, roft_pose_process_noise
//   CartesianQuaternionMeasurement (.h:27-123)               CartesianQuaternionMeasurement.h     (host-side mode machine)
//   UKFCorrection (UKFCorrection.h:24-52)                    UKFCorrection.h                      roft_ukf_correct
//   ROFTFilter (ROFTFilter.h:38-194)                         ROFTFilter.h                         roft_engine_* (one object)
//   CameraMeasurement, ImageSegmentationMeasurement,
//   ImageOpticalFlowSource, ImageOpticalFlowNVOF,
//   ModelParameters                                          Sources.h (+ a header per name)       roft_optical_flow (producer)
//   ImageSegmentationOFAidedSourceStamped<T>,
//   OpticalFlowQueueHandler                                  the same names                       roft_mask_propagate
//   DatasetImageOpticalFlow, DatasetImageSegmentation,
//   DatasetImageSegmentationDelayed, OpticalFlowUtilities,
//   MeshResource                                             the same names                       (files)
//
// Conventions kept from the reference: predict(prev, pred) / correct(pred, corr) on Gaussians; an invalid / empty
// measurement leaves corr = pred (SKFCorrection.cpp:46-69, UKFCorrection.cpp:64-68); constructors and unrecoverable errors
// throw std::runtime_error; freeze() returns a validity bool.  Header-only; link with libroft_hip.so.
#pragma once

#include "DatasetImageOpticalFlow.h"
#include "DatasetImageSegmentationDelayed.h"
#include "ImageSegmentationOFAidedSourceStamped.hpp"
#include "ROFTFilter.h"

namespace ROFT {

using compat::Gaussian;
using compat::MatrixXd;
using compat::VectorXd;

// ---- the whole tracker, batched: what the engine is built for ------------------------------------------------
// One instance tracks n objects; filtering_step() = ROFTFilter::filtering_step for all of them in the same launches.
class ROFTFilterBatch {
public:
    explicit ROFTFilterBatch(const roft_config& cfg) : cfg_(cfg)
    {
        compat::throw_if(roft_engine_create(&cfg_, &e_), "ROFTFilterBatch::ctor");
    }
    ~ROFTFilterBatch() { roft_engine_destroy(e_); }
    ROFTFilterBatch(const ROFTFilterBatch&) = delete;
    ROFTFilterBatch& operator=(const ROFTFilterBatch&) = delete;
    int add_object(const roft_object_desc& d)
    {
        int id = -1;
        compat::throw_if(roft_object_add(e_, &d, &id), "ROFTFilterBatch::add_object");
        return id;
    }
    // one frame: inputs[object]
    void filtering_step(const std::vector<roft_frame_input>& inputs)
    {
        compat::throw_if(roft_frame_submit(e_, inputs.data(), static_cast<int>(inputs.size())), "ROFTFilterBatch::filtering_step");
        compat::throw_if(roft_step(e_), "ROFTFilterBatch::filtering_step");
    }
    // a batch of consecutive frames: inputs[frame * n_objects + object]
    void filtering_steps(const std::vector<roft_frame_input>& inputs, int n_objects, int n_frames)
    {
        compat::throw_if(roft_frames_submit(e_, inputs.data(), n_objects, n_frames), "ROFTFilterBatch::filtering_steps");
        compat::throw_if(roft_step(e_), "ROFTFilterBatch::filtering_steps");
    }
    void wait() { compat::throw_if(roft_sync(e_), "ROFTFilterBatch::wait"); }
    void state(int obj, double pose13[13], double twist6[6])
    {
        compat::throw_if(roft_get_state(e_, obj, pose13, nullptr, twist6, nullptr), "ROFTFilterBatch::state");
    }
    roft_engine* engine() { return e_; }

private:
    roft_config cfg_;
    roft_engine* e_ = nullptr;
};

}  // namespace ROFT
