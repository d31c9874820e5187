// SPDX-License-Identifier: MIT
pragma solidity ^0.8.20;

import "./WithdrawVerifier.sol";

/// @title The proof-carrying replacement of `Owshen._processWithdraw`
/// @notice The reference gate (/root/reference/contracts/src/Owshen.sol:66-78) checks an owner signature over
///         keccak256(abi.encode(msg.sender, _tokenAddress, _amount, _id, block.chainid)) and a replay map `isExecuted[_id]`.
///         With proofs, every one of those five values is a public input of the statement (oracle/py/withdraw.py):
///           msg.sender   -> input[2] recipient         _tokenAddress -> input[4] token (also inside the leaf)
///           _amount      -> input[3] amount            block.chainid -> input[5] chain_id
///           _id          -> input[1] nullifier_hash (the replay map becomes `nullified`)
///         and input[0] is a commitment-tree root the sequencer has posted.  Nothing is a free parameter: a proof made for one
///         token, amount, recipient or chain does not verify for another (tests/withdraw_cases.py runs this gate's
///         model on proofs from the library).
/// @dev    Meant to be inherited by (or pasted into) Owshen.sol: `withdrawToken` / `withdrawNative`
///         (/root/reference/contracts/src/Owshen.sol:38-57) keep their shape with `bytes _signature, uint256 _id` replaced
///         by `uint256[8] proof, uint256 root, uint256 nullifierHash`.  Plain text in this repo (no solc in the image).
abstract contract OwshenWithdrawGate {
    WithdrawVerifier public verifier;
    mapping(uint256 => bool) public knownRoot;   // roots of the MiMC7 commitment tree, posted by the sequencer
    mapping(uint256 => bool) public nullified;   // replaces isExecuted[_id]

    /// BN254 scalar field modulus r (the reference's `Fp`, /root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:7-11)
    uint256 internal constant R = 21888242871839275222246405745257275088548364400416034343698204186575808495617;

    event WithdrawExecuted(address indexed to, address indexed token, uint256 nullifierHash, uint256 amount);

    function _processWithdraw(uint256[8] calldata proof, uint256 root, uint256 nullifierHash, address tokenAddress, uint256 amount)
        internal
    {
        // Canonical encodings only.  `Fp::from_repr` rejects bytes >= r, and so does this gate: x and x + r are the same field
        // element, so without the check `nullified[nullifierHash + r]` would be a second, unspent key for a spent note if the
        // verifier ever reduced its inputs (this one refuses them too -- the gate does not rely on that).  recipient and token are
        // `address` values: below 2^160 by type, far below r.  amount: the ledger's note amounts are field elements.
        require(root < R && nullifierHash < R, "ERROR: public input is not a field element.");
        require(amount < R, "ERROR: amount is not a field element.");
        require(knownRoot[root], "ERROR: unknown commitment root.");
        require(!nullified[nullifierHash], "ERROR: withdraw already executed.");
        uint256[6] memory input =
            [root, nullifierHash, uint256(uint160(msg.sender)), amount, uint256(uint160(tokenAddress)), block.chainid];
        require(verifier.verifyProof(proof, input), "ERROR: invalid proof.");
        nullified[nullifierHash] = true;
        emit WithdrawExecuted(msg.sender, tokenAddress, nullifierHash, amount);
    }
}
