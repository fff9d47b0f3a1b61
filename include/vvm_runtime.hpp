// vvm_runtime.hpp -- the reference's C++ runtime interface, spelled over the C ABI of libvvm_hip.so (vvm_abi.h).
//
// A caller written against Runtime/VectorVM/Public/VectorVMRuntime.h:10-33 (VectorVM::Runtime::AllocVectorVMState /
// FreeVectorVMState / ExecVectorVMState, FVectorVMExecContext) and VectorVM.h:246-313 (FDataSetMeta) compiles against this
// header with the same names, member names and argument meaning.  Nothing here computes: every type has the layout of its
// vvm_abi.h counterpart (checked below), and the three functions forward.  What differs from the reference is stated in
// vvm_abi.h: register streams and ID-table payloads are device memory, external functions are built-in ids.
//
// The engine's containers are not available outside it, so the two this interface uses are given here as far as the interface
// needs them: TArrayView (pointer + count, Core/Public/Containers/ArrayView.h:826-827) and, for the ID tables, TArray<int32>
// reduced to what the runtime reads of it (GetData / Num / Max).
#pragma once
#include <cstddef>
#include <cstdint>
#include <type_traits>

#include "vvm_abi.h"

namespace VectorVM
{
namespace Runtime
{
	using uint8 = std::uint8_t;
	using int32 = std::int32_t;
	using uint32 = std::uint32_t;
	constexpr int32 INDEX_NONE = -1;

	template <typename ElementType>
	class TArrayView
	{
	public:
		TArrayView() = default;
		TArrayView(ElementType* InData, int32 InCount) : DataPtr(InData), ArrayNum(InCount) {}
		template <std::size_t N> TArrayView(ElementType (&Array)[N]) : DataPtr(Array), ArrayNum((int32)N) {}
		ElementType* GetData() const { return DataPtr; }
		int32 Num() const { return ArrayNum; }
		ElementType& operator[](int32 Index) const { return DataPtr[Index]; }
		ElementType* begin() const { return DataPtr; }
		ElementType* end() const { return DataPtr + ArrayNum; }

	private:
		ElementType* DataPtr = nullptr;
		int32 ArrayNum = 0;
	};
	template <typename ElementType> using TConstArrayView = TArrayView<const ElementType>;

	// TArray<int32> as the runtime sees an ID table: payload in device memory, element count, capacity.
	struct FIDTable : vvm_id_table
	{
		FIDTable() { data = nullptr; num = 0; max = 0; }
		FIDTable(int32* DevicePayload, int32 InNum, int32 InMax) { data = DevicePayload; num = InNum; max = InMax; }
		int32* GetData() const { return data; }
		int32 Num() const { return num; }
		int32 Max() const { return max; }
	};

	// VectorVM.h:246-313
	struct FDataSetMeta
	{
		using FInputRegisterView = TArrayView<uint8 const* const>;
		using FOutputRegisterView = TArrayView<uint8* const>;

		FInputRegisterView InputRegisters;
		FOutputRegisterView OutputRegisters;
		uint32 InputRegisterTypeOffsets[3] = {0, 0, 0};
		uint32 OutputRegisterTypeOffsets[3] = {0, 0, 0};
		int32 DataSetAccessIndex = INDEX_NONE;
		int32 InstanceOffset = INDEX_NONE;
		FIDTable* IDTable = nullptr;
		FIDTable* FreeIDTable = nullptr;
		FIDTable* SpawnedIDsTable = nullptr;
		int32* NumFreeIDs = nullptr;
		int32* MaxUsedID = nullptr;
		int32* NumSpawnedIDs = nullptr;
		int32 IDAcquireTag = INDEX_NONE;

		FDataSetMeta() = default;
		FDataSetMeta(const FDataSetMeta&) = delete;
		FDataSetMeta& operator=(const FDataSetMeta&) = delete;

		void Reset()
		{
			InputRegisters = FInputRegisterView();
			OutputRegisters = FOutputRegisterView();
			DataSetAccessIndex = InstanceOffset = IDAcquireTag = INDEX_NONE;
			IDTable = FreeIDTable = SpawnedIDsTable = nullptr;
			NumFreeIDs = MaxUsedID = nullptr;
		}
		void Init(FInputRegisterView InInputRegisters, FOutputRegisterView InOutputRegisters, int32 InInstanceOffset, FIDTable* InIDTable,
		          FIDTable* InFreeIDTable, int32* InNumFreeIDs, int32* InNumSpawnedIDs, int32* InMaxUsedID, int32 InIDAcquireTag,
		          FIDTable* InSpawnedIDsTable)
		{
			InputRegisters = InInputRegisters;
			OutputRegisters = InOutputRegisters;
			DataSetAccessIndex = INDEX_NONE;
			InstanceOffset = InInstanceOffset;
			IDTable = InIDTable;
			FreeIDTable = InFreeIDTable;
			SpawnedIDsTable = InSpawnedIDsTable;
			NumFreeIDs = InNumFreeIDs;
			NumSpawnedIDs = InNumSpawnedIDs;
			MaxUsedID = InMaxUsedID;
			IDAcquireTag = InIDAcquireTag;
		}
	};

	// stands in for the delegate type of VectorVM.h:358: a built-in id (VVM_EXTFN_*)
	using FVMExternalFunction = vvm_ext_function;

	using FVectorVMState = vvm_state;

	// VectorVMRuntime.h:10-28
	struct FVectorVMExecContext
	{
		struct
		{
			uint32 NumBytesRequiredPerBatch;
			uint32 PerBatchRegisterDataBytesRequired;
			uint32 MaxChunksPerBatch;
			uint32 MaxInstancesPerChunk;
		} Internal = {0, 0, 0, 0};
		FVectorVMState* VVMState = nullptr;
		TArrayView<FDataSetMeta> DataSets;
		TArrayView<const FVMExternalFunction*> ExtFunctionTable;
		TArrayView<void*> UserPtrTable;
		int32 NumInstances = 0;
		const uint8* const* ConstantTableData = nullptr;
		const int* ConstantTableNumBytes = nullptr;
		int32 ConstantTableCount = 0;
	};

	// The forwarding below reinterprets these types as their C counterparts: same size, same member offsets.
#define VVM_SAME_MEMBER(CppType, CType, Member) \
	static_assert(offsetof(CppType, Member) == offsetof(CType, Member) && sizeof(((CppType*)nullptr)->Member) == sizeof(((CType*)nullptr)->Member), #Member)
	static_assert(std::is_standard_layout<FDataSetMeta>::value && sizeof(FDataSetMeta) == sizeof(vvm_dataset_meta), "FDataSetMeta");
	VVM_SAME_MEMBER(FDataSetMeta, vvm_dataset_meta, InputRegisters);
	VVM_SAME_MEMBER(FDataSetMeta, vvm_dataset_meta, OutputRegisters);
	VVM_SAME_MEMBER(FDataSetMeta, vvm_dataset_meta, InputRegisterTypeOffsets);
	VVM_SAME_MEMBER(FDataSetMeta, vvm_dataset_meta, OutputRegisterTypeOffsets);
	VVM_SAME_MEMBER(FDataSetMeta, vvm_dataset_meta, DataSetAccessIndex);
	VVM_SAME_MEMBER(FDataSetMeta, vvm_dataset_meta, InstanceOffset);
	VVM_SAME_MEMBER(FDataSetMeta, vvm_dataset_meta, IDTable);
	VVM_SAME_MEMBER(FDataSetMeta, vvm_dataset_meta, FreeIDTable);
	VVM_SAME_MEMBER(FDataSetMeta, vvm_dataset_meta, SpawnedIDsTable);
	VVM_SAME_MEMBER(FDataSetMeta, vvm_dataset_meta, NumFreeIDs);
	VVM_SAME_MEMBER(FDataSetMeta, vvm_dataset_meta, MaxUsedID);
	VVM_SAME_MEMBER(FDataSetMeta, vvm_dataset_meta, NumSpawnedIDs);
	VVM_SAME_MEMBER(FDataSetMeta, vvm_dataset_meta, IDAcquireTag);
	static_assert(std::is_standard_layout<FVectorVMExecContext>::value && sizeof(FVectorVMExecContext) == sizeof(vvm_exec_context), "FVectorVMExecContext");
	VVM_SAME_MEMBER(FVectorVMExecContext, vvm_exec_context, Internal);
	VVM_SAME_MEMBER(FVectorVMExecContext, vvm_exec_context, VVMState);
	VVM_SAME_MEMBER(FVectorVMExecContext, vvm_exec_context, DataSets);
	VVM_SAME_MEMBER(FVectorVMExecContext, vvm_exec_context, ExtFunctionTable);
	VVM_SAME_MEMBER(FVectorVMExecContext, vvm_exec_context, UserPtrTable);
	VVM_SAME_MEMBER(FVectorVMExecContext, vvm_exec_context, NumInstances);
	VVM_SAME_MEMBER(FVectorVMExecContext, vvm_exec_context, ConstantTableData);
	VVM_SAME_MEMBER(FVectorVMExecContext, vvm_exec_context, ConstantTableNumBytes);
	VVM_SAME_MEMBER(FVectorVMExecContext, vvm_exec_context, ConstantTableCount);
	static_assert(sizeof(TArrayView<void*>) == 16 && sizeof(FIDTable) == sizeof(vvm_id_table), "container layout");
#undef VVM_SAME_MEMBER

	// VectorVMRuntime.h:30-33.  nullptr on an empty or malformed blob, as the reference (VectorVMRuntime.cpp:2313-2316).
	inline FVectorVMState* AllocVectorVMState(TConstArrayView<uint8> ContextData)
	{
		return vvm_alloc_state(ContextData.GetData(), (std::size_t)ContextData.Num());
	}
	inline void FreeVectorVMState(FVectorVMState* State) { vvm_free_state(State); }
	// void like the reference, which returns silently where this library refuses (VectorVMRuntime.cpp:2184-2187, 2264-2267):
	// DataSetAccessIndex is then left as it was; LastExecStatus() / vvm_last_error() tell why.
	inline int& LastExecStatusWord() { static thread_local int Status = VVM_OK; return Status; }
	inline void ExecVectorVMState(FVectorVMExecContext* ExecCtx)
	{
		LastExecStatusWord() = vvm_exec(reinterpret_cast<vvm_exec_context*>(ExecCtx));
	}
	inline int LastExecStatus() { return LastExecStatusWord(); }
}  // namespace Runtime
}  // namespace VectorVM
