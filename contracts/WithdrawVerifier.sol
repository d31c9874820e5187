// SPDX-License-Identifier: MIT
pragma solidity ^0.8.20;

/// @title Groth16 verifier for the Owshen withdraw circuit (BN254, EIP-196 / EIP-197 precompiles)
/// @notice The step after the GPU prover path (SURVEY.md 8f-2).  The reference contract authorises a withdrawal with an
///         owner signature (`Owshen._processWithdraw`, /root/reference/contracts/src/Owshen.sol:66-78: keccak + ECDSA/1271 +
///         replay map); this verifier is what that gate calls instead once withdrawals carry a proof (see
///         contracts/README.md for the patched `_processWithdraw`).
/// @dev    Layout follows the snarkjs verifier convention: G2 coordinates are passed as (x.c1, x.c0, y.c1, y.c0), all
///         words are 32-byte big-endian.  `owshen_amd/evm.py` emits the 28 verifying-key words from an "OWVK0001" blob
///         (constructor argument) and the 8 proof words from a 256-byte proof (`proof_to_evm_calldata`).
///         Statement: public inputs = (root, nullifier_hash, recipient, amount, token, chain_id) -- oracle/py/withdraw.py:
///         everything the ECDSA gate signed (msg.sender, token, amount, id, chainid), the nullifier hash standing in for `id`.
contract WithdrawVerifier {
    uint256 internal constant Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583; // base field
    uint256 internal constant R = 21888242871839275222246405745257275088548364400416034343698204186575808495617; // scalar field
    uint256 internal constant N_PUB = 6;

    // vk[0..1] alpha (G1) | vk[2..5] beta (G2) | vk[6..9] gamma (G2) | vk[10..13] delta (G2) | vk[14 + 2 i ..] IC_i (G1), i = 0..6
    uint256[28] public vk;

    constructor(uint256[28] memory vkWords) {
        for (uint256 i = 0; i < 28; i++) {
            require(vkWords[i] < Q, "vk word not a field element");
            vk[i] = vkWords[i];
        }
    }

    /// @param proof  A.x, A.y, B.x.c1, B.x.c0, B.y.c1, B.y.c0, C.x, C.y
    /// @param input  root, nullifier_hash, recipient, amount, token, chain_id (each < R)
    /// @return ok    true iff  e(-A, B) e(alpha, beta) e(IC_0 + sum input_i IC_{i+1}, gamma) e(C, delta) == 1
    function verifyProof(uint256[8] calldata proof, uint256[6] memory input) public view returns (bool ok) {
        for (uint256 i = 0; i < 8; i++) {
            if (proof[i] >= Q) return false;
        }
        uint256[2] memory acc = [vk[14], vk[15]];
        for (uint256 i = 0; i < N_PUB; i++) {
            if (input[i] >= R) return false;
            uint256[2] memory term;
            bool success;
            uint256[3] memory mulIn = [vk[16 + 2 * i], vk[17 + 2 * i], input[i]];
            assembly {
                success := staticcall(gas(), 7, mulIn, 0x60, term, 0x40)
            }
            if (!success) return false;
            uint256[4] memory addIn = [acc[0], acc[1], term[0], term[1]];
            assembly {
                success := staticcall(gas(), 6, addIn, 0x80, acc, 0x40)
            }
            if (!success) return false;
        }
        uint256[24] memory p;
        // -A: (x, Q - y); the point at infinity (0, 0) stays (0, 0)
        p[0] = proof[0];
        p[1] = (proof[0] == 0 && proof[1] == 0) ? 0 : (Q - proof[1]) % Q;
        p[2] = proof[2];
        p[3] = proof[3];
        p[4] = proof[4];
        p[5] = proof[5];
        p[6] = vk[0];
        p[7] = vk[1];
        p[8] = vk[2];
        p[9] = vk[3];
        p[10] = vk[4];
        p[11] = vk[5];
        p[12] = acc[0];
        p[13] = acc[1];
        p[14] = vk[6];
        p[15] = vk[7];
        p[16] = vk[8];
        p[17] = vk[9];
        p[18] = proof[6];
        p[19] = proof[7];
        p[20] = vk[10];
        p[21] = vk[11];
        p[22] = vk[12];
        p[23] = vk[13];
        uint256[1] memory out;
        bool done;
        assembly {
            done := staticcall(gas(), 8, p, 0x300, out, 0x20)
        }
        return done && out[0] == 1;
    }
}
