// ppenv_ta_chain.h — interface between ppenv_ta_sim.hip (the C ABI of the 27-DoF step) and ppenv_ta_chain.hip (its chain-wave kernel).
#pragma once

#include "ppenv_ta_device.h"

namespace pp {
namespace ta {

// everything one launch of the chain-wave step needs besides the model scalars (ppenv_ta_step's arguments)
struct TAChainArgs {
    ppenv_ta_params p;
    const StepConsts* K;          // device copy of the scene constants (read by the ball wave only)
    const float* actions;         // [N,27]
    const float* initial_rb;      // [N,42,13]
    float* root_states;           // [N,3,13]   in / out
    float* dof_states;            // [N,27,2]   in / out
    float* rb_states;             // [N,42,13]  out, or NULL: not materialised
    float* dof_force;             // [N,27]     out
    float* pre_vx;                // [N]        out, or NULL
    const float* reset_override;  // [N,5] or NULL
    uint32_t* flags;
    uint32_t* episode;
    long long* progress;
    float* obs;                   // [N,313]
    float* rew;
    long long* reset;
    uint32_t* scratch;            // one word, zero between launches: bit 0 = some env reset in this launch, bits 1.. = workgroups done
    uint32_t* status;             // host-visible status word of the handle (bit 0: a hand-off timed out)
    // optional second copy of the observation rows, as the policy's first layer wants them (ppenv_ta_sim_set_policy_input; NULL: off)
    const float* pin_mean;        // [313]
    const float* pin_inv_std;     // [313]
    float pin_clip;
    unsigned short* pin_out;      // fp16 [N, pin_ld], columns >= 313 zero
    int pin_ld;
    // domain randomisation (ppenv_ta_sim_set_randomization; every pointer NULL and both sigmas 0: off, the plain kernel is launched)
    const float* dr_kp;           // [27][N] drive stiffness scales
    const float* dr_kd;           // [27][N] drive damping scales
    const float* dr_ms;           // [28][N] link mass scales (link 0 = pelvis; mass and inertia together)
    const float* dr_es;           // [N] restitution scale of the humanoid's shapes and the paddle
    const float* dr_fs;           // [N] friction scale of the same
    float dr_act_sigma, dr_obs_sigma;
    bool dr_on() const { return dr_kp || dr_kd || dr_ms || dr_es || dr_fs || dr_act_sigma > 0.f || dr_obs_sigma > 0.f; }
};

// does the run-time model equal, bit for bit, the tables the chain-wave kernel was compiled from?
bool ta_chain_model_matches(const TAConsts& C, char* why = nullptr, size_t nwhy = 0);
// enqueue one step; returns a PPENV_* code
int ta_chain_launch(const TAScal& P, const TAChainArgs& a, void* stream);

}  // namespace ta
}  // namespace pp
